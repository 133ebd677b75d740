// ddG head, sequence-logit epilogue, sequence embedding and the derived-table preparation (gfx950).
//
// Reference semantics: TransferModel.forward (/root/reference/transfer_model.py:86-120) evaluated once
// per POSITION instead of once per mutation; LightAttention on a length-1 sequence (:148-155) = centre
// tap of feature_convolution (softmax over a size-1 axis is 1, dropout is identity in eval);
// both_out = [ReLU, Linear] x3 (:67-71); ddg_out = Linear(1,1) (:73); W_out + log_softmax
// (protein_mpnn_utils.py:1275-1276); W_s embedding (:1238).
#include <stdlib.h>

#include "tmpnn_common.h"
#include "tmpnn_internal.h"
#include "tmpnn_split.h"
#include "tmpnn_head_body.h"

template <int NRB>   // tile = 16*NRB residues (chosen by the launcher for load balance / small batches)
__global__ __launch_bounds__(TM_THREADS, 1) void head_kernel(HeadArgs a) {
    constexpr int ROWS = 16 * NRB;
    __shared__ __attribute__((aligned(16))) float tX[3][ROWS * TM_H];
    __shared__ __attribute__((aligned(16))) float tY[3][ROWS * TM_H];
    __shared__ int s_S[ROWS];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int n_tiles = (a.T + ROWS - 1) / ROWS;
    const float dw = a.ddg_w[0], db = a.ddg_b[0];

    for (int tile = tm_bid(); tile < n_tiles; tile += tm_nblk()) {
        const int r0 = tile * ROWS, rows = min(ROWS, a.T - r0);
        if (tid < ROWS) s_S[tid] = tid < rows ? a.S[r0 + tid] : 0;
        if (a.maxlen_probe && a.status && tid < rows && a.maxlen_probe[(size_t)(r0 + tid) * TM_KS] < 0) atomicOr(a.status, TMPNN_STATUS_MAXLEN);
        load_tile<NRB>(tX[0], a.hA + (size_t)r0 * TM_H, rows, tid);
        load_tile<NRB>(tX[1], a.hB + (size_t)r0 * TM_H, rows, tid);
        if (a.status) {     // the ReLUs of both_out map NaN to 0: a poisoned decoder state must be flagged at the input
            bool bad = false;
            const unsigned *ua = reinterpret_cast<const unsigned *>(a.hA), *ub = reinterpret_cast<const unsigned *>(a.hB);
            for (int k = tid; k < rows * TM_H; k += TM_THREADS)
                bad |= tm_nonfinite_bits(ua[(size_t)r0 * TM_H + k]) || tm_nonfinite_bits(ub[(size_t)r0 * TM_H + k]);
            if (bad) atomicOr(a.status, TMPNN_STATUS_RANGE);
        }
#pragma unroll
        for (int it = 0; it < 2 * NRB; ++it) {  // x[256:384] = W_s[S]
            const int idx = it * TM_THREADS + tid, row = idx >> 5, c = idx & 31;
            const int s = row < rows ? a.S[r0 + row] : 0;
            st4(tX[2] + chunk_off(row, c), ld4(a.Ws + s * TM_H + 4 * c));
        }
        __syncthreads();

        // y = relu(Wc x + bc), 384 -> 384 in three 128-column groups
        for (int g = 0; g < 3; ++g) {
            f4 acc[NRB][2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const f4 b = ld4(a.conv_b + 128 * g + 32 * wv + 16 * cb + 4 * q);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[rb][cb] = b;
            }
            for (int kt = 0; kt < 3; ++kt) {
                float wf[2][32];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    load_wfrag<8>(a.conv_center, 384, 128 * g + 32 * wv + 16 * cb, 128 * kt, 384, wf[cb], lane);
                mma_tile<8, 2, 128, NRB>(tX[kt], wf, acc, lane);
            }
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    st4(tY[g] + chunk_off(16 * rb + m, 8 * wv + 4 * cb + q), relu4(acc[rb][cb]));
        }
        __syncthreads();

        {   // 384 -> 64, relu; wavefront w owns columns 16w..16w+15 -> tX[0][:, 0:64]
            f4 acc[NRB][1];
            const f4 b = ld4(a.b1 + 16 * wv + 4 * q);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b;
            for (int kt = 0; kt < 3; ++kt) {
                float wf[1][32];
                load_wfrag<8>(a.w1, 384, 16 * wv, 128 * kt, 64, wf[0], lane);
                mma_tile<8, 1, 128, NRB>(tY[kt], wf, acc, lane);
            }
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) st4(tX[0] + chunk_off(16 * rb + m, 4 * wv + q), relu4(acc[rb][0]));
        }
        __syncthreads();
        if (wv < 2) {   // 64 -> 32, relu -> tX[1][:, 0:32]
            f4 acc[NRB][1];
            const f4 b = ld4(a.b2 + 16 * wv + 4 * q);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b;
            float wf[1][16];
            load_wfrag<4>(a.w2, 64, 16 * wv, 0, 32, wf[0], lane);
            mma_tile<4, 1, 128, NRB>(tX[0], wf, acc, lane);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) st4(tX[1] + chunk_off(16 * rb + m, 4 * wv + q), relu4(acc[rb][0]));
        }
        __syncthreads();
        if (wv < 2) {   // 32 -> 21 (rows 21..31 of the weight read as zero) -> z in tX[2][:, 0:32]
            f4 acc[NRB][1];
            f4 b;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * wv + 4 * q + r;
                b[r] = n < TMPNN_VOCAB ? a.b3[n] : 0.f;
            }
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b;
            float wf[1][8];
            load_wfrag<2>(a.w3, 32, 16 * wv, 0, TMPNN_VOCAB, wf[0], lane);
            mma_tile<2, 1, 128, NRB>(tX[1], wf, acc, lane);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) st4(tX[2] + chunk_off(16 * rb + m, 4 * wv + q), acc[rb][0]);
        }
        __syncthreads();
        for (int e = tid; e < rows * TMPNN_VOCAB; e += TM_THREADS) {
            const int row = e / TMPNN_VOCAB, aa = e - row * TMPNN_VOCAB;
            const float z = tX[2][chunk_off(row, aa >> 2) + (aa & 3)];
            const int wt = s_S[row];
            const float zw = tX[2][chunk_off(row, wt >> 2) + (wt & 3)];
            const float dd = (dw * z + db) - (dw * zw + db);   // :110-116
            a.ddg[(size_t)(r0 + row) * TMPNN_VOCAB + aa] = dd;
            if (a.status && tm_nonfinite(dd)) atomicOr(a.status, TMPNN_STATUS_RANGE);
            if (a.z_opt) a.z_opt[(size_t)(r0 + row) * TMPNN_VOCAB + aa] = z;
        }
        __syncthreads();
    }
}

template <typename SP, int NRB, bool IMG>
__global__ __launch_bounds__(512, 2) void head8_split_kernel(HeadArgs a) { head8_body<SP, NRB, IMG>(a); }

// log_softmax(W_out h + b): one wavefront per residue, lane a < 21 owns logit a.
__global__ __launch_bounds__(TM_THREADS) void log_probs_kernel(const float *__restrict__ W, const float *__restrict__ b,
                                                               const float *__restrict__ h, int T,
                                                               float *__restrict__ out, int32_t *__restrict__ status,
                                                               const int32_t *__restrict__ maxlen_probe) {
    const int lane = tm_tid() & 63, wv = tm_tid() >> 6;
    for (int t = tm_bid() * 4 + wv; t < T; t += tm_nblk() * 4) {
        float logit = -INFINITY;
        if (maxlen_probe && status && lane == 0 && maxlen_probe[(size_t)t * TM_KS] < 0) atomicOr(status, TMPNN_STATUS_MAXLEN);   // see HeadArgs
        if (status) {      // poisoned input row -> flag (raw-bit test on the loaded values, see tm_nonfinite_bits)
            const unsigned *ur = reinterpret_cast<const unsigned *>(h + (size_t)t * TM_H);
            if (tm_nonfinite_bits(ur[lane]) || tm_nonfinite_bits(ur[lane + 64])) atomicOr(status, TMPNN_STATUS_RANGE);
        }
        if (lane < TMPNN_VOCAB) {
            const float *wr = W + lane * TM_H, *hr = h + (size_t)t * TM_H;
            float s = 0.f;
#pragma unroll 8
            for (int k = 0; k < TM_H; ++k) s += wr[k] * hr[k];
            logit = s + b[lane];
        }
        float mx = logit;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float e = lane < TMPNN_VOCAB ? expf(logit - mx) : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off);
        if (lane < TMPNN_VOCAB) {
            const float lp = (logit - mx) - logf(e);
            out[(size_t)t * TMPNN_VOCAB + lane] = lp;
            if (status && tm_nonfinite(lp)) atomicOr(status, TMPNN_STATUS_RANGE);
        }
    }
}

__global__ __launch_bounds__(TM_THREADS) void seq_embed_kernel(const float *__restrict__ Ws, const int32_t *__restrict__ S,
                                                               int64_t T, float *__restrict__ hS) {
    const int64_t total = T * 32, stride = (int64_t)tm_nblk() * TM_THREADS;
    for (int64_t g = (int64_t)tm_bid() * TM_THREADS + tm_tid(); g < total; g += stride) {
        const int64_t t = g >> 5;
        const int c = (int)(g & 31);
        st4(hS + g * 4, ld4(Ws + S[t] * TM_H + 4 * c));
    }
}

// derived tables, computed once per weight set
__global__ void prep_pos_table_kernel(const float *__restrict__ pos_w, const float *__restrict__ pos_b,
                                      const float *__restrict__ edge_w, float *__restrict__ table) {
    const int d = tm_bid(), n = tm_tid();     // 66 x 128
    float s = 0.f;
    for (int p = 0; p < 16; ++p) s += (pos_w[p * 66 + d] + pos_b[p]) * edge_w[n * 416 + p];
    table[d * TM_H + n] = s;
}
__global__ void prep_seq_table_kernel(const float *__restrict__ Ws, const float *__restrict__ W1,
                                      float *__restrict__ table) {
    const int s = tm_bid(), n = tm_tid();     // 21 x 128;  W1 [128,512], columns 256..383 multiply W_s[S_j]
    float acc = 0.f;
    for (int k = 0; k < TM_H; ++k) acc += Ws[s * TM_H + k] * W1[n * 512 + 256 + k];
    table[s * TM_H + n] = acc;
}
__global__ void prep_conv_center_kernel(const float *__restrict__ conv_w, float *__restrict__ center) {
    const int i = tm_bid() * tm_bdim() + tm_tid();   // 384*384
    if (i < 384 * 384) center[i] = conv_w[(size_t)i * 9 + 4];
}

int launch_prep_tables(tmpnn_weights *w, hipStream_t st) {
    prep_pos_table_kernel<<<66, TM_H, 0, st>>>(w->pos_w, w->pos_b, w->edge_w, w->pos_table);
    for (int l = 0; l < 3; ++l)
        prep_seq_table_kernel<<<TMPNN_VOCAB, TM_H, 0, st>>>(w->Ws_w, w->dec[l].W1, w->seq_table[l]);
    if (w->n_tensors == TMPNN_N_TENSORS)
        prep_conv_center_kernel<<<(384 * 384 + 255) / 256, 256, 0, st>>>(w->conv_w, w->conv_center);
    return tm_check_launch("prep_tables");
}

HeadArgs tm_head_args(const tmpnn_weights *w, const float *hA, const float *hB, const int32_t *S, int64_t T, float *ddg, float *z_opt,
                      int32_t *status, const int32_t *maxlen_probe) {
    HeadArgs a{w->conv_center, w->conv_b, w->mlp_w[0], w->mlp_b[0], w->mlp_w[1], w->mlp_b[1], w->mlp_w[2], w->mlp_b[2],
               w->ddg_w, w->ddg_b, w->Ws_w, hA, hB, S, ddg, z_opt, (int)T, status, maxlen_probe, {}};
    bool have_img = tm_matmul_mode() == TM_MM_F16X2;
    for (int u = 0; u < 12 && have_img; ++u) {
        a.img[u] = tm_find_wimg(u < 9 ? w->conv_center + (size_t)128 * (u / 3) * 384 + 128 * (u % 3) : w->mlp_w[0] + 128 * (u - 9));
        have_img = a.img[u] != nullptr;
    }
    if (!have_img) for (int u = 0; u < 12; ++u) a.img[u] = nullptr;
    return a;
}

int launch_head(const tmpnn_weights *w, const float *hA, const float *hB, const int32_t *S, int64_t T, float *ddg,
                float *z_opt, int32_t *status, hipStream_t st, const int32_t *maxlen_probe) {
    HeadArgs a = tm_head_args(w, hA, hB, S, T, ddg, z_opt, status, maxlen_probe);
    // tile height for load balance, as in node_update: 1 workgroup per CU, ~1.2 MB of weights streamed per tile
    const int64_t slots = tm_num_cus();
    int best_rows = 48;
    int64_t best_cost = -1;
    for (int rows = 48; rows >= 16; rows -= 16) {
        const int64_t tiles = (T + rows - 1) / rows, rounds = (tiles + slots - 1) / slots;
        const int64_t cost = rounds * (rows + 16);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_rows = rows; }
    }
    const int64_t tiles = (T + best_rows - 1) / best_rows;
    const int grid = (int)(tiles < slots ? tiles : slots);
    tm_prof_begin("head", st);
    static const bool split_ok = TM_DBG_FLAG("TMPNN_HEAD_SPLIT", true);
    if (tm_matmul_mode() == TM_MM_F16X2 && split_ok) {
#define TM_HEAD8(NRB)                                                             \
    if (a.img[0]) head8_split_kernel<SplitH2, NRB, true><<<grid, 512, 0, st>>>(a); \
    else head8_split_kernel<SplitH2, NRB, false><<<grid, 512, 0, st>>>(a)
        if (best_rows == 16) { TM_HEAD8(1); }
        else if (best_rows == 32) { TM_HEAD8(2); }
        else { TM_HEAD8(3); }
#undef TM_HEAD8
        tm_prof_end(st);
        return tm_check_launch("ddg_head");
    }
    if (best_rows == 16) head_kernel<1><<<grid, TM_THREADS, 0, st>>>(a);
    else if (best_rows == 32) head_kernel<2><<<grid, TM_THREADS, 0, st>>>(a);
    else head_kernel<3><<<grid, TM_THREADS, 0, st>>>(a);
    tm_prof_end(st);
    return tm_check_launch("ddg_head");
}

int launch_log_probs(const tmpnn_weights *w, const float *h, int64_t T, float *out, int32_t *status, hipStream_t st,
                     const int32_t *maxlen_probe) {
    const int64_t blocks = (T + 3) / 4, cap = (int64_t)tm_num_cus() * 8;
    { tm_prof_begin("log_probs", st); log_probs_kernel<<<(int)(blocks < cap ? blocks : cap), TM_THREADS, 0, st>>>(w->Wout_w, w->Wout_b, h, (int)T, out, status, maxlen_probe); tm_prof_end(st); }
    return tm_check_launch("log_probs");
}

// A forward that returns hidden states only (no ddG head, no log-probabilities) has no kernel that looks at them: this one does.
__global__ __launch_bounds__(TM_THREADS) void range_check_kernel(const unsigned *__restrict__ x, int64_t n4, int32_t *__restrict__ status,
                                                                 const int32_t *__restrict__ maxlen_probe, int64_t T) {
    typedef unsigned uv4 __attribute__((ext_vector_type(4)));
    bool bad = false;
    if (maxlen_probe) {                                          // see HeadArgs::maxlen_probe
        bool longer = false;
        for (int64_t t = (int64_t)tm_bid() * TM_THREADS + tm_tid(); t < T; t += (int64_t)tm_nblk() * TM_THREADS)
            longer = longer || maxlen_probe[t * TM_KS] < 0;
        if (longer) atomicOr(status, TMPNN_STATUS_MAXLEN);
    }
    for (int64_t i = (int64_t)tm_bid() * TM_THREADS + tm_tid(); i < n4; i += (int64_t)tm_nblk() * TM_THREADS) {
        const uv4 v = reinterpret_cast<const uv4 *>(x)[i];
        bad = bad || tm_nonfinite_bits(v.x) || tm_nonfinite_bits(v.y) || tm_nonfinite_bits(v.z) || tm_nonfinite_bits(v.w);
    }
    if (bad) atomicOr(status, TMPNN_STATUS_RANGE);
}

int launch_range_check(const float *x, int64_t n, int32_t *status, hipStream_t st, const int32_t *maxlen_probe, int64_t T) {
    if (!status || n <= 0) return TMPNN_OK;
    const int64_t n4 = n / 4, blocks = (n4 + TM_THREADS - 1) / TM_THREADS, cap = (int64_t)tm_num_cus() * 4;     // n is a multiple of 128 here
    range_check_kernel<<<(int)(blocks < cap ? blocks : cap), TM_THREADS, 0, st>>>(reinterpret_cast<const unsigned *>(x), n4, status, maxlen_probe, T);
    return tm_check_launch("range_check");
}

int launch_seq_embed(const tmpnn_weights *w, const int32_t *S, int64_t T, float *hS, hipStream_t st) {
    const int64_t blocks = (T * 32 + TM_THREADS - 1) / TM_THREADS, cap = (int64_t)tm_num_cus() * 8;
    { tm_prof_begin("seq_embed", st); seq_embed_kernel<<<(int)(blocks < cap ? blocks : cap), TM_THREADS, 0, st>>>(w->Ws_w, S, T, hS); tm_prof_end(st); }
    return tm_check_launch("seq_embed");
}

// ------------------------------------------------------------------------------------------------
// Generic head: any TransferModel configuration the reference constructor accepts (transfer_model.py:45-73) —
// num_final_layers 0..3 decoder states in the input, LightAttention on or off (:106-108), any list of hidden_dims.
// The released configuration (2 states, LightAttention, [64, 32]) runs head8_split_kernel above; this path keeps retrained
// heads runnable: x = [h_dec(last) | ... | W_s[S]] -> (centre tap of feature_convolution) -> [ReLU, Linear] x n -> ddG.
// One dense kernel on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, weights as the A operand so a lane ends up with four
// consecutive output columns of one row), operands straight from global memory: sized for correctness and generality.
// ------------------------------------------------------------------------------------------------
struct ConcatArgs { const float *hid[3]; int n_final; const float *Ws; const int32_t *S; float *X; int T, D0; int32_t *status; };

__global__ __launch_bounds__(TM_THREADS) void head_concat_kernel(ConcatArgs a) {
    const int per_row = a.D0 / 4;
    bool bad = false;
    for (int64_t e = (int64_t)tm_bid() * TM_THREADS + tm_tid(); e < (int64_t)a.T * per_row; e += (int64_t)tm_nblk() * TM_THREADS) {
        const int t = (int)(e / per_row), c = (int)(e - (int64_t)t * per_row), blk = c / 32, c4 = c - blk * 32;
        typedef unsigned uv4 __attribute__((ext_vector_type(4)));
        uv4 v;
        if (blk < a.n_final) {
            v = *reinterpret_cast<const uv4 *>(a.hid[blk] + (size_t)t * TM_H + 4 * c4);
            // the ReLU in front of the first Linear maps NaN to 0: a poisoned decoder state is flagged here, on the raw bits
            bad = bad || tm_nonfinite_bits(v.x) || tm_nonfinite_bits(v.y) || tm_nonfinite_bits(v.z) || tm_nonfinite_bits(v.w);
        } else {
            v = *reinterpret_cast<const uv4 *>(a.Ws + (size_t)a.S[t] * TM_H + 4 * c4);
        }
        *reinterpret_cast<uv4 *>(a.X + (size_t)t * a.D0 + 4 * c) = v;
    }
    if (bad && a.status) atomicOr(a.status, TMPNN_STATUS_RANGE);
}

// Y[t, n] = b[n] + sum_k act(X[t, k]) W[n, k]; W element (n, k) at W[n * ldw + wk0 + k * wks] (a Linear: wks 1; the centre tap of a
// [N, K, 9] convolution weight: ldw 9 K, wks 9, wk0 4). One wavefront per 16-row tile, all column blocks in turn.
struct DenseArgs { const float *X; const float *W; const float *b; float *Y; int T, K, N, ldw, wks, wk0, relu_in; };

__global__ __launch_bounds__(TM_THREADS) void dense_generic_kernel(DenseArgs a) {
    const int lane = tm_tid() & 63, wv = tm_wave(tm_tid()), m = lane & 15, q = lane >> 4;
    const int n_tiles = (a.T + 15) / 16;
    for (int tile = tm_bid() * 4 + wv; tile < n_tiles; tile += tm_nblk() * 4) {
        const int row = tile * 16 + m;
        const bool row_ok = row < a.T;
        const float *x = a.X + (size_t)(row_ok ? row : 0) * a.K;
        for (int n0 = 0; n0 < a.N; n0 += 16) {
            const int wrow = n0 + m;
            const bool w_ok = wrow < a.N;
            const float *w = a.W + (size_t)(w_ok ? wrow : 0) * a.ldw + a.wk0;
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < a.K; k += 4) {
                const int kk = k + q;
                const bool k_ok = kk < a.K;
                float xv = row_ok && k_ok ? x[kk] : 0.f;
                if (a.relu_in) xv = fmaxf(xv, 0.f);
                const float wvv = w_ok && k_ok ? w[(size_t)kk * a.wks] : 0.f;
                acc = mfma16(wvv, xv, acc);
            }
            if (row_ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = n0 + 4 * q + r;
                    if (col < a.N) a.Y[(size_t)row * a.N + col] = acc[r] + a.b[col];
                }
            }
        }
    }
}

// ddg[t, a] = (w z[t, a] + b) - (w z[t, S[t]] + b) (:110-116; subtract_mut) — or w z + b alone; column 20 ('X') included like the
// fused head's table.
__global__ __launch_bounds__(TM_THREADS) void ddg_from_z_kernel(const float *__restrict__ z, const int32_t *__restrict__ S,
                                                                const float *__restrict__ dwp, const float *__restrict__ dbp, int T,
                                                                float *__restrict__ ddg, int32_t *__restrict__ status) {
    const float dw = dwp[0], db = dbp[0];
    for (int64_t e = (int64_t)tm_bid() * TM_THREADS + tm_tid(); e < (int64_t)T * TMPNN_VOCAB; e += (int64_t)tm_nblk() * TM_THREADS) {
        const int t = (int)(e / TMPNN_VOCAB);
        const float dd = (dw * z[e] + db) - (dw * z[(size_t)t * TMPNN_VOCAB + S[t]] + db);
        ddg[e] = dd;
        if (status && tm_nonfinite(dd)) atomicOr(status, TMPNN_STATUS_RANGE);
    }
}

int launch_head_generic(const float *const *hidden, int n_final, const float *Ws, const int32_t *S, int64_t T, const float *conv_w,
                        const float *conv_b, int n_layers, const float *const *mlp_w, const float *const *mlp_b, const int32_t *dims,
                        const float *ddg_w, const float *ddg_b, float *ddg, float *z_opt, float *buf0, float *buf1, int32_t *status,
                        hipStream_t st) {
    const int D0 = dims[0];
    const int64_t cap = (int64_t)tm_num_cus() * 8;
    auto grid_for = [&](int64_t work_items) { const int64_t b = (work_items + TM_THREADS - 1) / TM_THREADS; return (int)(b < cap ? (b > 0 ? b : 1) : cap); };
    ConcatArgs c{{n_final > 0 ? hidden[0] : nullptr, n_final > 1 ? hidden[1] : nullptr, n_final > 2 ? hidden[2] : nullptr}, n_final, Ws, S, buf0,
                 (int)T, D0, status};
    tm_prof_begin("head", st);
    head_concat_kernel<<<grid_for(T * (D0 / 4)), TM_THREADS, 0, st>>>(c);
    float *cur = buf0, *nxt = buf1;
    const int64_t tiles = (T + 15) / 16;
    const int dgrid = (int)((tiles + 3) / 4 < cap ? (tiles + 3) / 4 : cap);
    if (conv_w) {       // LightAttention on a length-1 sequence = the centre tap (index 4 of 9) of feature_convolution + its bias
        DenseArgs d{cur, conv_w, conv_b, nxt, (int)T, D0, D0, 9 * D0, 9, 4, 0};
        dense_generic_kernel<<<dgrid, TM_THREADS, 0, st>>>(d);
        std::swap(cur, nxt);
    }
    for (int l = 0; l < n_layers; ++l) {
        const bool last = l == n_layers - 1;
        float *out = last && z_opt ? z_opt : nxt;
        DenseArgs d{cur, mlp_w[l], mlp_b[l], out, (int)T, dims[l], dims[l + 1], dims[l], 1, 0, 1};     // both_out: ReLU FIRST (:69-71)
        dense_generic_kernel<<<dgrid, TM_THREADS, 0, st>>>(d);
        if (out == nxt) std::swap(cur, nxt); else cur = out;
    }
    ddg_from_z_kernel<<<grid_for(T * TMPNN_VOCAB), TM_THREADS, 0, st>>>(cur, S, ddg_w, ddg_b, (int)T, ddg, status);
    tm_prof_end(st);
    return tm_check_launch("ddg_head_generic");
}
