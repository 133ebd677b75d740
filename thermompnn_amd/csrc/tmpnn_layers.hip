// Message-passing layers of the ProteinMPNN encoder/decoder as hand-written gfx950 kernels.
//
// Reference semantics: EncLayer.forward (/root/reference/protein_mpnn_utils.py:816-839),
// DecLayer.forward (:859-880), decoder wiring in ProteinMPNN.forward (:1238-1273).
//
// Schedule (parity-neutral restructuring, checked on CPU by tests/test_schedule_model.py):
//   W1 . [h_i | e_ij | h_j]  =  (W1a h_i + b1) + W1b e_ij + W1c h_j      node terms once per NODE (node_proj)
//   sum_k m_k (W3 x_k + b3)  =  W3 (sum_k m_k x_k) + b3 sum_k m_k        W3 once per NODE (node_update)
// so the per-EDGE work is two (message) or three (edge update) 128x128 GEMMs on the matrix cores, with
// the weight slices resident in VGPRs for the whole kernel and the residue's 48-slot neighbour list +
// edge tile staged in LDS. Per-node aggregation over K is an in-workgroup column reduction
// (deterministic; no atomics).
#include <stdlib.h>

#include "tmpnn_common.h"
#include "tmpnn_internal.h"

// ------------------------------------------------------------------------------------------------
// node_proj: P[t, 0:128] = Wa h_t + ba ; P[t, 128:256] = Wc h_t            (48 residues per tile)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TM_THREADS, 2) void node_proj_kernel(const float *__restrict__ h,
                                                                  const float *__restrict__ Wa, int lda,
                                                                  const float *__restrict__ ba,
                                                                  const float *__restrict__ Wc, int ldc, int T,
                                                                  float *__restrict__ P,
                                                                  const float *__restrict__ add_tab,
                                                                  const int32_t *__restrict__ add_idx) {
    __shared__ __attribute__((aligned(16))) float tA[TM_TILE * TM_H];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    float wa[2][32], wc[2][32];
    f4 bias[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int n0 = 32 * wv + 16 * cb;
        load_wfrag<8>(Wa, lda, n0, 0, TM_H, wa[cb], lane);
        load_wfrag<8>(Wc, ldc, n0, 0, TM_H, wc[cb], lane);
        bias[cb] = ld4(ba + n0 + 4 * q);
    }
    const int n_tiles = (T + TM_TILE - 1) / TM_TILE;
    for (int tile = tm_bid(); tile < n_tiles; tile += tm_nblk()) {
        const int r0 = tile * TM_TILE;
        load_tile(tA, h + (size_t)r0 * TM_H, min(TM_TILE, T - r0), tid);
        __syncthreads();
        f4 acc[3][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {      // A half (with bias), then C half
#pragma unroll
            for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = half ? f4{0.f, 0.f, 0.f, 0.f} : bias[cb];
            if (half) mma_tile<8, 2>(tA, wc, acc, lane);
            else mma_tile<8, 2>(tA, wa, acc, lane);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int row = r0 + 16 * rb + m;
                if (row < T) {
                    const float *add = half && add_tab ? add_tab + add_idx[row] * TM_H : nullptr;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const int n = 32 * wv + 16 * cb + 4 * q;
                        st4(P + (size_t)row * 256 + 128 * half + n, add ? ld4(add + n) + acc[rb][cb] : acc[rb][cb]);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// msg: per residue i (tile = its 48 neighbour slots):  Ssum_i = sum_k ma_ik * gelu(W2 gelu(pre_ik) + b2)
//   encoder: pre = A_i + C_j + W1b e_ij                  ma = mask_i mask_j   (EncLayer :819-825, :1232-1233)
//   decoder: pre = A_i + mask_i (W1b e_ij + SeqT[S_j] + D_j)   ma = 1         (DecLayer :863-870, :1270-1272)
//            (SeqT[S_j] + D_j arrives pre-added in the neighbour half of P: NodeProj::add_tab)
// ------------------------------------------------------------------------------------------------
struct MsgArgs {
    const float *W1e; int ld1;
    const float *W2, *b2;
    const float *P;          // [T,256]: A (bias folded) | C or D (+ sequence term)
    const float *hE;         // [T,48,128]
    const int32_t *E_idx;    // [T,48]
    const float *mask;       // [T]
    float *Ssum, *cnt;
    int T;
    int skew;                // start delay of the second half of the grid, in units of 64 cycles
};

// NW wavefronts per workgroup, two workgroups per CU either way: NW = 4 -> 32 columns per wavefront (128 weight VGPRs,
// 2 wavefronts per SIMD); NW = 8 -> 16 columns (64 weight VGPRs, <= 128 VGPRs in total, 4 wavefronts per SIMD).
template <bool DEC, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void msg_kernel(MsgArgs a) {
    constexpr int NT = 64 * NW, NCB = 8 / NW;
    __shared__ __attribute__((aligned(16))) float tE[TM_TILE * TM_H];
    __shared__ __attribute__((aligned(16))) float tA[TM_TILE * TM_H];
    __shared__ float s_part[3][TM_H];
    __shared__ int s_idx[TM_TILE];
    __shared__ float s_ma[TM_TILE];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int col0 = (TM_H / NW) * wv, chunk0 = (32 / NW) * wv;

    float w1[NCB][32], w2[NCB][32];
    f4 bias2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int n0 = col0 + 16 * cb;
        load_wfrag<8>(a.W1e, a.ld1, n0, 0, TM_H, w1[cb], lane);
        load_wfrag<8>(a.W2, TM_H, n0, 0, TM_H, w2[cb], lane);
        bias2[cb] = ld4(a.b2 + n0 + 4 * q);
    }

    const TileRange tr = xcd_tile_range(a.T);
    for (int i = tr.begin; i < tr.end; i += tr.step) {
        const float mi = a.mask[i];
        if (tid < TM_TILE) {
            const int j = a.E_idx[(size_t)i * TM_KS + tid];
            s_idx[tid] = j;
            s_ma[tid] = j < 0 ? 0.f : (DEC ? 1.f : mi * a.mask[j]);
        }
        load_tile_async<NW>(tE, a.hE + (size_t)i * TM_KS * TM_H, wv, lane);
        __syncthreads();

        f4 acc[3][NCB];
        int jrow[3];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int j = s_idx[16 * rb + m];
            jrow[rb] = j < 0 ? i : j;
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int n = col0 + 16 * cb + 4 * q;
            if (DEC) {   // the gathered terms ride in the accumulator: their L2 latency hides under the GEMM
#pragma unroll
                for (int rb = 0; rb < 3; ++rb)
                    acc[rb][cb] = ld4(a.P + (size_t)jrow[rb] * 256 + 128 + n);
            } else {
                const f4 ai = ld4(a.P + (size_t)i * 256 + n);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) acc[rb][cb] = ai + ld4(a.P + (size_t)jrow[rb] * 256 + 128 + n);
            }
        }
        mma_tile<8, NCB>(tE, w1, acc, lane);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int n = col0 + 16 * cb + 4 * q;
            f4 ai;
            if (DEC) ai = ld4(a.P + (size_t)i * 256 + n);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                f4 v = acc[rb][cb];
                if (DEC) v = ai + mi * v;
                st4(tA + chunk_off(16 * rb + m, chunk0 + 4 * cb + q), gelu4(v));
            }
        }
        __syncthreads();

#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = bias2[cb];
        mma_tile<8, NCB>(tA, w2, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const float ma = s_ma[16 * rb + m];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                f4 v = gelu4(acc[rb][cb]) * ma;
                if (ma == 0.f) v = f4{0.f, 0.f, 0.f, 0.f};
                st4(tE + chunk_off(16 * rb + m, chunk0 + 4 * cb + q), v);
            }
        }
        __syncthreads();

        // per-node aggregation over the 48 slots: column sums over NT/128 row groups, combined in a fixed order
        {
            constexpr int G = NT / TM_H, RPG = TM_TILE / G;       // groups, rows per group
            const int n = tid & 127, grp = tid >> 7;
            float s = 0.f;
#pragma unroll 8
            for (int r = RPG * grp; r < RPG * grp + RPG; ++r) s += tE[chunk_off(r, n >> 2) + (n & 3)];
            if (grp) s_part[grp - 1][n] = s;
            __syncthreads();
            if (!grp) {
#pragma unroll
                for (int g = 0; g < G - 1; ++g) s += s_part[g][n];
                a.Ssum[(size_t)i * TM_H + n] = s;
            }
            if (tid == 128) {
                float c = 0.f;
                for (int r = 0; r < TM_TILE; ++r) c += s_ma[r];
                a.cnt[i] = c;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// enc_edge: e_ij <- LN3(e_ij + W13 gelu(W12 gelu(A'_i + C'_j + W11b e_ij) + b12) + b13)   (EncLayer :834-838)
// ------------------------------------------------------------------------------------------------
struct EdgeArgs {
    const float *W11e;  // W11[:, 128:256], ld 384
    const float *W12, *b12, *W13, *b13, *g3, *be3;
    const float *P;     // [T,256]
    float *hE;
    const int32_t *E_idx;
    int T;
};

// ------------------------------------------------------------------------------------------------
// enc_edge, 8-wavefront form: 512 threads, wavefront w owns ONE 16-column block (96 weight VGPRs instead of 192), so
// two wavefronts share each SIMD: the matrix pipe sees the same 192 MFMAs per GEMM per SIMD, but their issue, the
// LDS waits and above all the VALU phases (GELU, LayerNorm) interleave between two instruction streams — a single
// wavefront issues only about one instruction every 4-5 cycles (MI355X_MICROARCH.md, per-instruction constants).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void row_stats_partial1(const f4 v, float *stat_slot, int q) {
    float mean = (v.x + v.y + v.z + v.w) * 0.25f;
    const f4 d = v - mean;
    float m2 = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    float n = 4.f;
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const float mo = __shfl_xor(mean, off), m2o = __shfl_xor(m2, off);
        const float delta = mo - mean;
        mean = 0.5f * (mean + mo);
        m2 = m2 + m2o + delta * delta * (0.5f * n);
        n *= 2.f;
    }
    if (q == 0) { stat_slot[0] = mean; stat_slot[1] = m2; }
}
// merge eight per-wavefront (mean, M2) partials of 16 columns each
__device__ __forceinline__ void row_stats_finish8(const float *stat_row /* 16 floats */, float &mean, float &rstd) {
    f4 p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = ld4(stat_row + 4 * k);
    mean = 0.125f * (p[0].x + p[0].z + p[1].x + p[1].z + p[2].x + p[2].z + p[3].x + p[3].z);
    float m2 = 0.f, dd = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float d0 = p[k].x - mean, d1 = p[k].z - mean;
        m2 += p[k].y + p[k].w;
        dd += d0 * d0 + d1 * d1;
    }
    rstd = 1.0f / sqrtf((m2 + 16.f * dd) * (1.0f / 128.0f) + 1e-5f);
}

__global__ __launch_bounds__(512, 2) void enc_edge8_kernel(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) float tE[2][TM_TILE * TM_H];
    __shared__ __attribute__((aligned(16))) float tA[TM_TILE * TM_H];
    __shared__ __attribute__((aligned(16))) float tB[TM_TILE * TM_H];
    __shared__ __attribute__((aligned(16))) float s_stat[TM_TILE][16];
    __shared__ int s_idx[2][TM_TILE];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;

    float w11[1][32], w12[1][32], w13[1][32];
    load_wfrag<8>(a.W11e, 384, 16 * wv, 0, TM_H, w11[0], lane);
    load_wfrag<8>(a.W12, TM_H, 16 * wv, 0, TM_H, w12[0], lane);
    load_wfrag<8>(a.W13, TM_H, 16 * wv, 0, TM_H, w13[0], lane);
    const int ncol = 16 * wv + 4 * q, chunk = 4 * wv + q;
    const f4 b12 = ld4(a.b12 + ncol), b13 = ld4(a.b13 + ncol);
    const int c32 = lane & 31;
    const f4 g4 = ld4(a.g3 + 4 * c32), be4 = ld4(a.be3 + 4 * c32);

    const TileRange tr = xcd_tile_range(a.T);
    int i = tr.begin;
    int cur = 0;
    f4 gai, gcj[3];
    if (i < tr.end) {
        if (tid < TM_TILE) s_idx[0][tid] = a.E_idx[(size_t)i * TM_KS + tid];
        load_tile_async<8>(tE[0], a.hE + (size_t)i * TM_KS * TM_H, wv, lane);
        __syncthreads();
        gai = ld4(a.P + (size_t)i * 256 + ncol);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int j = s_idx[0][16 * rb + m];
            gcj[rb] = ld4(a.P + (size_t)(j < 0 ? i : j) * 256 + 128 + ncol);
        }
    }
    for (; i < tr.end; i += tr.step) {
        float *tile_g = a.hE + (size_t)i * TM_KS * TM_H;
        const float *tEc = tE[cur];
        const int inext = i + tr.step;
        const bool has_next = inext < tr.end;
        int nidx = -1;
        if (has_next) {
            load_tile_async<8>(tE[cur ^ 1], a.hE + (size_t)inext * TM_KS * TM_H, wv, lane);
            if (tid < TM_TILE) nidx = a.E_idx[(size_t)inext * TM_KS + tid];
        }
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = gai + gcj[rb];
        mma_tile<8, 1>(tEc, w11, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) st4(tA + chunk_off(16 * rb + m, chunk), gelu4(acc[rb][0]));
        if (has_next && tid < TM_TILE) s_idx[cur ^ 1][tid] = nidx;
        __syncthreads();

        if (has_next) {
            gai = ld4(a.P + (size_t)inext * 256 + ncol);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int j = s_idx[cur ^ 1][16 * rb + m];
                gcj[rb] = ld4(a.P + (size_t)(j < 0 ? inext : j) * 256 + 128 + ncol);
            }
        }
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b12;
        mma_tile<8, 1>(tA, w12, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) st4(tB + chunk_off(16 * rb + m, chunk), gelu4(acc[rb][0]));
        __syncthreads();

#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b13;
        mma_tile<8, 1>(tB, w13, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int off = chunk_off(16 * rb + m, chunk);
            const f4 v = ld4(tEc + off) + acc[rb][0];          // residual
            st4(tA + off, v);
            row_stats_partial1(v, &s_stat[16 * rb + m][2 * wv], q);
        }
        __syncthreads();

#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int row = 6 * wv + 2 * it + (lane >> 5);
            float mean, rstd;
            row_stats_finish8(&s_stat[row][0], mean, rstd);
            const f4 y = (ld4(tA + chunk_off(row, c32)) - mean) * rstd * g4 + be4;
            if (s_idx[cur][row] >= 0) st4(tile_g + (size_t)row * TM_H + 4 * c32, y);
        }
        cur ^= 1;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// node_update:  h <- mask * LN2(h1 + FFN(h1)),  h1 = LN1(h + (W3 Ssum + cnt b3) / 30)
//   (EncLayer :826-832 / DecLayer :870-879; PositionWiseFeedForward :883-893)
// followed, in the same kernel, by up to two node projections of the NEW state (what node_proj computes):
//   P_k[t, 0:128] = Wa_k h_t + ba_k ; P_k[t, 128:256] = Wc_k h_t
// (k = 0: the edge update of this encoder layer, k = 1: the message pass of the NEXT layer), which removes two
// launches and two re-reads of h per layer. Tile = 16*NRB residues: 48 for batches, 16 when T is small so that a
// single protein still spreads over more CUs (the weights stream from L2 either way).
// ------------------------------------------------------------------------------------------------

template <int NRB>
__global__ __launch_bounds__(TM_THREADS, 2) void node_update_kernel(NodeArgs a) {
    constexpr int ROWS = 16 * NRB;
    __shared__ __attribute__((aligned(16))) float tA[ROWS * TM_H];
    __shared__ __attribute__((aligned(16))) float tB[ROWS * TM_H];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int c32 = lane & 31;
    const int n_tiles = (a.T + ROWS - 1) / ROWS;

    for (int tile = tm_bid(); tile < n_tiles; tile += tm_nblk()) {
        const int r0 = tile * ROWS;
        load_tile<NRB>(tA, a.Ssum + (size_t)r0 * TM_H, min(ROWS, a.T - r0), tid);
        __syncthreads();

        float wf[2][32];
        f4 acc[NRB][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) load_wfrag<8>(a.W3, TM_H, 32 * wv + 16 * cb, 0, TM_H, wf[cb], lane);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f4{0.f, 0.f, 0.f, 0.f};
        mma_tile<8, 2, 128, NRB>(tA, wf, acc, lane);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int row = r0 + 16 * rb + m;
            const bool ok = row < a.T;
            const float c = ok ? a.cnt[row] : 0.f;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int n = 32 * wv + 16 * cb + 4 * q;
                const f4 hv = ok ? ld4(a.h_in + (size_t)row * TM_H + n) : f4{0.f, 0.f, 0.f, 0.f};
                const f4 dh = (acc[rb][cb] + c * ld4(a.b3 + n)) / 30.0f;
                st4(tB + chunk_off(16 * rb + m, 8 * wv + 4 * cb + q), hv + dh);
            }
        }
        __syncthreads();
        {   // LN1 in place
            const f4 g4 = ld4(a.n1w + 4 * c32), b4 = ld4(a.n1b + 4 * c32);
#pragma unroll
            for (int it = 0; it < 2 * NRB; ++it) {
                const int row = 4 * NRB * wv + 2 * it + (lane >> 5);
                float *p = tB + chunk_off(row, c32);
                st4(p, layer_norm_row(ld4(p), g4, b4));
            }
        }
        __syncthreads();

        f4 out[NRB][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const f4 b = ld4(a.bout + 32 * wv + 16 * cb + 4 * q);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) out[rb][cb] = b;
        }
        for (int c = 0; c < 4; ++c) {           // FFN hidden 512 in four 128-wide chunks
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int n0 = 128 * c + 32 * wv + 16 * cb;
                load_wfrag<8>(a.Win, TM_H, n0, 0, 512, wf[cb], lane);
                const f4 b = ld4(a.bin + n0 + 4 * q);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[rb][cb] = b;
            }
            mma_tile<8, 2, 128, NRB>(tB, wf, acc, lane);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    st4(tA + chunk_off(16 * rb + m, 8 * wv + 4 * cb + q), gelu4(acc[rb][cb]));
            __syncthreads();
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) load_wfrag<8>(a.Wout, 512, 32 * wv + 16 * cb, 128 * c, TM_H, wf[cb], lane);
            mma_tile<8, 2, 128, NRB>(tA, wf, out, lane);
            __syncthreads();
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int off = chunk_off(16 * rb + m, 8 * wv + 4 * cb + q);
                st4(tA + off, ld4(tB + off) + out[rb][cb]);
            }
        __syncthreads();
        {   // LN2, mask, coalesced store; the new state also stays in tB for the projections
            const f4 g4 = ld4(a.n2w + 4 * c32), b4 = ld4(a.n2b + 4 * c32);
#pragma unroll
            for (int it = 0; it < 2 * NRB; ++it) {
                const int row = 4 * NRB * wv + 2 * it + (lane >> 5);
                const int grow = r0 + row;
                f4 y = layer_norm_row(ld4(tA + chunk_off(row, c32)), g4, b4);
                y = grow < a.T ? y * a.mask[grow] : f4{0.f, 0.f, 0.f, 0.f};
                st4(tB + chunk_off(row, c32), y);
                if (grow < a.T) st4(a.h_out + (size_t)grow * TM_H + 4 * c32, y);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const ProjSpec &ps = a.proj[k];
            if (ps.P == nullptr) continue;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int n0 = 32 * wv + 16 * cb;
                    if (half) load_wfrag<8>(ps.Wc, ps.ldc, n0, 0, TM_H, wf[cb], lane);
                    else load_wfrag<8>(ps.Wa, ps.lda, n0, 0, TM_H, wf[cb], lane);
                    const f4 b = half ? f4{0.f, 0.f, 0.f, 0.f} : ld4(ps.ba + n0 + 4 * q);
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) acc[rb][cb] = b;
                }
                mma_tile<8, 2, 128, NRB>(tB, wf, acc, lane);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const int row = r0 + 16 * rb + m;
                    if (row < a.T) {
                        const float *add = half && ps.add_tab ? ps.add_tab + ps.add_idx[row] * TM_H : nullptr;
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
                            const int n = 32 * wv + 16 * cb + 4 * q;
                            st4(ps.P + (size_t)row * 256 + 128 * half + n, add ? ld4(add + n) + acc[rb][cb] : acc[rb][cb]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static int grid_for(int64_t work_items, int blocks_per_cu) {
    const int64_t cap = (int64_t)tm_num_cus() * blocks_per_cu;
    return (int)(work_items < cap ? (work_items < 1 ? 1 : work_items) : cap);
}

int launch_node_proj(const float *h, const NodeProj &np, int64_t T, hipStream_t st) {
    const int64_t tiles = (T + TM_TILE - 1) / TM_TILE;
    { tm_prof_begin("node_proj", st); node_proj_kernel<<<grid_for(tiles, 2), TM_THREADS, 0, st>>>(h, np.Wa, np.lda, np.ba, np.Wc, np.ldc, (int)T, np.P, np.add_tab, np.add_idx); tm_prof_end(st); }
    return tm_check_launch("node_proj");
}

int launch_msg(bool dec, const float *W1e, int ld1, const float *W2, const float *b2, const float *P,
               const float *hE, const int32_t *E_idx, const float *mask, int64_t T, float *Ssum, float *cnt, hipStream_t st) {
    tm_prof_begin(dec ? "dec_msg" : "enc_msg", st);
    if (tm_matmul_mode() != TM_MM_FP32) {
        const int rc = launch_msg_split(tm_matmul_mode(), dec, W1e, ld1, W2, b2, P, hE, E_idx, mask, T, Ssum, cnt, st);
        tm_prof_end(st);
        return rc;
    }
    MsgArgs a{W1e, ld1, W2, b2, P, hE, E_idx, mask, Ssum, cnt, (int)T, 0};
    if (dec) msg_kernel<true, 4><<<grid_for(T, 2), 256, 0, st>>>(a);
    else msg_kernel<false, 4><<<grid_for(T, 2), 256, 0, st>>>(a);
    tm_prof_end(st);
    return tm_check_launch(dec ? "dec_msg" : "enc_msg");
}

int launch_enc_edge(const EncW &e, const float *P, float *hE, const int32_t *E_idx, int64_t T, hipStream_t st) {
    tm_prof_begin("enc_edge", st);
    if (tm_matmul_mode() != TM_MM_FP32) {        // split-precision 16-bit matrix-core forms (tmpnn_split.hip)
        const int rc = launch_enc_edge_split(tm_matmul_mode(), e, P, hE, E_idx, T, st);
        tm_prof_end(st);
        return rc;
    }
    EdgeArgs a{e.W11 + 128, e.W12, e.b12, e.W13, e.b13, e.norm3_w, e.norm3_b, P, hE, E_idx, (int)T};
    enc_edge8_kernel<<<grid_for(T, 1), 512, 0, st>>>(a);
    tm_prof_end(st);
    return tm_check_launch("enc_edge8");
}

int launch_node_update(const float *W3, const float *b3, const float *n1w, const float *n1b, const float *Win,
                       const float *bin, const float *Wout, const float *bout, const float *n2w, const float *n2b,
                       const float *h_in, const float *Ssum, const float *cnt, const float *mask, int64_t T,
                       float *h_out, const NodeProj *p0, const NodeProj *p1, hipStream_t st, const HeadArgs *head, bool *head_ran) {
    if (head_ran) *head_ran = false;
    NodeArgs a{W3, b3, n1w, n1b, Win, bin, Wout, bout, n2w, n2b, h_in, Ssum, cnt, mask, h_out, (int)T, {}};
    const NodeProj *ps[2] = {p0, p1};
    for (int k = 0; k < 2; ++k)
        a.proj[k] = ps[k] ? ProjSpec{ps[k]->Wa, ps[k]->lda, ps[k]->ba, ps[k]->Wc, ps[k]->ldc, ps[k]->P, ps[k]->add_tab, ps[k]->add_idx}
                          : ProjSpec{nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr};
    {   // pre-built fragment images (all 13 units or none)
        const float *base[13] = {W3};
        for (int c = 0; c < 4; ++c) { base[1 + 2 * c] = Win + (size_t)128 * c * 128; base[2 + 2 * c] = Wout + 128 * c; }
        for (int k = 0; k < 2; ++k) { base[9 + 2 * k] = ps[k] ? ps[k]->Wa : nullptr; base[10 + 2 * k] = ps[k] ? ps[k]->Wc : nullptr; }
        bool all = true;
        for (int u = 0; u < 13; ++u) {
            a.img[u] = base[u] ? tm_find_wimg(base[u]) : nullptr;
            if (base[u] && !a.img[u]) all = false;
        }
        static const bool img_on = TM_DBG_FLAG("TMPNN_NODE_IMG", true);
        if (!all || !img_on) for (int u = 0; u < 13; ++u) a.img[u] = nullptr;
    }
    tm_prof_begin("node_update", st);
    // Tile height (16 / 32 / 48 residues) chosen for load balance: the grid offers 2 workgroup slots per CU, every
    // tile streams the same ~0.8 MB of weights from L2 (worth about 16 rows of MFMA time), so minimise
    // rounds x (rows + 16). E.g. T = 16384: 48-row tiles = 342 tiles -> some CUs get 96 rows; 32-row tiles = exactly 512.
    const int64_t slots = (int64_t)2 * tm_num_cus();
    int best_rows = 48;
    int64_t best_cost = -1;
    for (int rows = 48; rows >= 16; rows -= 16) {
        const int64_t tiles = (T + rows - 1) / rows, rounds = (tiles + slots - 1) / slots;
        const int64_t cost = rounds * (rows + 16);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_rows = rows; }
    }
    static const bool split_ok = TM_DBG_FLAG("TMPNN_NODE_SPLIT", true);
    if (tm_matmul_mode() == TM_MM_F16X2 && split_ok) {
        const int rc = launch_node_update_split(a, T, st, head, head_ran);
        tm_prof_end(st);
        return rc;
    }
    const int64_t tiles = (T + best_rows - 1) / best_rows;
    if (best_rows == 16) node_update_kernel<1><<<grid_for(tiles, 2), TM_THREADS, 0, st>>>(a);
    else if (best_rows == 32) node_update_kernel<2><<<grid_for(tiles, 2), TM_THREADS, 0, st>>>(a);
    else node_update_kernel<3><<<grid_for(tiles, 2), TM_THREADS, 0, st>>>(a);
    tm_prof_end(st);
    return tm_check_launch("node_update");
}

// ------------------------------------------------------------------------------------------------
// clock probe (tools/clock_probe.py): every workgroup runs a dependent-free stream of fp32 MFMAs fed from LDS, like the
// GEMM phases of the edge kernels, and reports shader cycles (s_memtime) against the 100 MHz constant clock.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TM_THREADS, 1) void clock_probe_kernel(int iters, unsigned long long *out, float *sink) {
    __shared__ __attribute__((aligned(16))) float tile[TM_TILE * TM_H];
    const int tid = tm_tid(), lane = tid & 63;
    for (int k = tid; k < TM_TILE * TM_H; k += TM_THREADS) tile[k] = 1e-3f * (float)(k & 255);
    float w[2][32];
#pragma unroll
    for (int k = 0; k < 32; ++k) { w[0][k] = 1e-3f * (float)(k + lane); w[1][k] = 1e-3f * (float)(k - lane); }
    f4 acc[3][2];
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) mma_tile<8, 2>(tile, w, acc, lane);
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    float sum = 0.f;
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) sum += acc[rb][cb].x + acc[rb][cb].w;
    if (sum == 123.456f) sink[tid] = sum;
    if (tid == 0) { out[2 * tm_bid()] = c1 - c0; out[2 * tm_bid() + 1] = t1 - t0; }
}

int launch_clock_probe(int blocks, int iters, unsigned long long *out, float *sink, hipStream_t st) {
    clock_probe_kernel<<<blocks, TM_THREADS, 0, st>>>(iters, out, sink);
    return tm_check_launch("clock_probe");
}

// One sleeping wavefront (no LDS, a handful of registers: it fits beside any kernel of the library) that reads the shader cycle counter
// and the 100 MHz reference before and after `iters` x s_sleep 127: launched on a side stream while the forward runs on another, it
// reports the clock the chip actually keeps UNDER THAT LOAD (round 5: 2.03 GHz inside the f16x2 message kernels against 2.38 GHz
// with their MFMAs removed — the pipeline runs power-limited, docs/NOTEBOOK.md 9.10).
__global__ void clock_monitor_kernel(int iters, unsigned long long *out) {
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(127);
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    if (tm_tid() == 0) { out[0] = c1 - c0; out[1] = t1 - t0; }
}
int launch_clock_monitor(int iters, unsigned long long *out, hipStream_t st) {
    clock_monitor_kernel<<<1, 64, 0, st>>>(iters, out);
    return tm_check_launch("clock_monitor");
}
