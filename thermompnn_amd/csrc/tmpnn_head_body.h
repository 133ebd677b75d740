// The ddG head's argument block and the body of its 8-wavefront f16x2 kernel, shared by tmpnn_head.hip (head8_split_kernel) and
// tmpnn_node.hip (node_head_fused_kernel). Reference: TransferModel.forward, /root/reference/transfer_model.py:86-120.
#pragma once
#include "tmpnn_common.h"
#include "tmpnn_internal.h"
#include "tmpnn_split.h"

struct HeadArgs {
    const float *conv_center, *conv_b;   // [384,384], [384]
    const float *w1, *b1, *w2, *b2, *w3, *b3;   // 384->64, 64->32, 32->21
    const float *ddg_w, *ddg_b;
    const float *Ws;                      // [21,128]
    const float *hA, *hB;                 // last / previous decoder state [T,128]
    const int32_t *S;
    float *ddg, *z_opt;
    int T;
    int32_t *status;                      // may be null: TMPNN_STATUS_RANGE is OR-ed in when a ddG is not finite
    const int32_t *maxlen_probe;          // fused forward: E_idx [T,48]; slot 0 < 0 marks a row the k-NN kernel left empty because its
                                          // protein is longer than max_len -> TMPNN_STATUS_MAXLEN (the k-NN kernel zeroes the word and
                                          // therefore cannot OR into it itself: no memset launch in front of the forward)
    const char *img[12];                  // f16 fragment images of the 12 GEMM units (WImg, tmpnn_internal.h) or all null
};

__device__ __forceinline__ f4 relu4(f4 v) { return f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }

// ------------------------------------------------------------------------------------------------
// head, 8-wavefront f16x2 form (default in f16x2 mode): the 384 -> 384 centre-tap GEMM and the 384 -> 64 layer (12 GEMM
// units of K = 128) on the 16-bit matrix cores, 16 output columns per wavefront, the fp32 weight fragment of unit u+1
// fetched from L2 under the MFMAs of unit u (as in node_update8_split_kernel); the two tiny layers (64 -> 32 -> 21) and
// the ddG epilogue are the fp32 code of head_kernel.
// ------------------------------------------------------------------------------------------------
// The kernel's body as a device function (round 6): head8_split_kernel runs it on its own, node_head_fused_kernel (tmpnn_node.hip) behind
// the last decoder layer's node update of the same 16 residues (small launches: one launch less per forward).
template <typename SP, int NRB, bool IMG>
__device__ __forceinline__ void head8_body(const HeadArgs &a) {
    constexpr int ROWS = 16 * NRB, PLT = SP::NP * ROWS * 256;
    static_assert(PLT >= ROWS * TM_H * 4, "fp32 tiles of the small layers are aliased on dead x planes");
    __shared__ __attribute__((aligned(16))) char pX[3][PLT];
    __shared__ __attribute__((aligned(16))) char pY[3][PLT];
    __shared__ int s_S[ROWS];
    float *tF0 = reinterpret_cast<float *>(pX[0]), *tF1 = reinterpret_cast<float *>(pX[1]), *tF2 = reinterpret_cast<float *>(pX[2]);
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const int n_tiles = (a.T + ROWS - 1) / ROWS;
    const float dw = a.ddg_w[0], db = a.ddg_b[0];

    // units 0..8: conv_center rows 128 g + 16 wv + m, columns 128 kt (u = 3 g + kt); units 9..11: w1 rows 16 wv + m (wv < 4)
    auto src = [&](int u) -> const float * {
        if (u < 9) return a.conv_center + (size_t)(128 * (u / 3) + 16 * wv + m) * 384 + 128 * (u % 3) + 8 * q;
        return a.w1 + (size_t)(16 * (wv & 3) + m) * 384 + 128 * (u - 9) + 8 * q;
    };
    f4 raw[8];
    auto issue = [&](int u) {
        if constexpr (IMG) {                                     // ready-made planes, 8 coalesced loads (see node_update8_split_kernel)
            const char *p = a.img[u] + (size_t)wv * 8192 + lane * 16;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                raw[2 * c] = *reinterpret_cast<const f4 *>(p + 2048 * c);
                raw[2 * c + 1] = *reinterpret_cast<const f4 *>(p + 2048 * c + 1024);
            }
            return;
        }
        const float *p = src(u);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            raw[2 * c] = ld4(p + 32 * c);
            raw[2 * c + 1] = ld4(p + 32 * c + 4);
        }
    };
    WFragS<SP> wf[1][4];
    auto split_raw = [&]() {
        if constexpr (IMG) {
            static_assert(SP::NP == 2, "the fragment images hold the two f16x2 planes");
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                wf[0][c].p[0] = __builtin_bit_cast(u4, raw[2 * c]);
                wf[0][c].p[1] = __builtin_bit_cast(u4, raw[2 * c + 1]);
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned w4[4][SP::NP];
            SP::split2(f2{raw[2 * c].x, raw[2 * c].y}, w4[0]);
            SP::split2(f2{raw[2 * c].z, raw[2 * c].w}, w4[1]);
            SP::split2(f2{raw[2 * c + 1].x, raw[2 * c + 1].y}, w4[2]);
            SP::split2(f2{raw[2 * c + 1].z, raw[2 * c + 1].w}, w4[3]);
#pragma unroll
            for (int p = 0; p < SP::NP; ++p) wf[0][c].p[p] = u4{w4[0][p], w4[1][p], w4[2][p], w4[3][p]};
        }
    };

    // Small operands of every layer once per workgroup, BEFORE the first weight-fragment request (gfx9 retires loads in order:
    // requested inside the tile, each of these cost its own L2 round trip in front of the MFMAs that needed it — the two tiny
    // layers' weights after a barrier, the biases at every accumulator initialisation).
    f4 cb[3], b1v, b2v, b3v;
#pragma unroll
    for (int g = 0; g < 3; ++g) cb[g] = ld4(a.conv_b + 128 * g + ncol);
    b1v = ld4(a.b1 + (ncol & 63));
    b2v = ld4(a.b2 + (ncol & 31));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = (ncol & 31) + r;
        b3v[r] = a.b3[n < TMPNN_VOCAB ? n : 0];
        if (n >= TMPNN_VOCAB) b3v[r] = 0.f;
    }
    float w32[1][16], w8[1][8];
    load_wfrag<4>(a.w2, 64, 16 * (wv & 1), 0, 32, w32[0], lane);
    load_wfrag<2>(a.w3, 32, 16 * (wv & 1), 0, TMPNN_VOCAB, w8[0], lane);

    int tile = tm_bid();
    if (tile < n_tiles) issue(0);
    for (; tile < n_tiles; tile += tm_nblk()) {
        const int r0 = tile * ROWS, rows = min(ROWS, a.T - r0);
        if (tid < ROWS) s_S[tid] = tid < rows ? a.S[r0 + tid] : 0;
        if (a.maxlen_probe && a.status && tid < rows && a.maxlen_probe[(size_t)(r0 + tid) * TM_KS] < 0) atomicOr(a.status, TMPNN_STATUS_MAXLEN);
        for (int idx = tid; idx < ROWS * 32; idx += 512) {      // x = [h_last | h_prev | W_s[S]] -> planes
            const int row = idx >> 5, c = idx & 31;
            const bool ok = row < rows;
            const size_t grow = (size_t)(r0 + (ok ? row : 0));   // rows past T: a valid row, masked below (no branch around the loads)
            // the ReLUs of both_out map NaN to 0, so a poisoned decoder state would come out as a finite ddG: flag it here, on
            // the raw bits as loaded (see tm_nonfinite_bits)
            typedef unsigned uv4 __attribute__((ext_vector_type(4)));
            const uv4 zero4 = uv4{0u, 0u, 0u, 0u};
            const int sres = a.S[grow];
            uv4 ra = *reinterpret_cast<const uv4 *>(a.hA + grow * TM_H + 4 * c);
            uv4 rb = *reinterpret_cast<const uv4 *>(a.hB + grow * TM_H + 4 * c);
            const f4 vs = ld4(a.Ws + (ok ? sres : 0) * TM_H + 4 * c);
            if (!ok) { ra = zero4; rb = zero4; }
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) bad = bad || tm_f16_range_bits(ra[k]) || tm_f16_range_bits(rb[k]);   // non-finite, or finite but beyond fp16: the planes below could not carry it
            if (a.status && bad) atomicOr(a.status, TMPNN_STATUS_RANGE);
            const f4 va = __builtin_bit_cast(f4, ra), vb = __builtin_bit_cast(f4, rb);
            store_split<SP, ROWS>(pX[0], row, c, va);
            store_split<SP, ROWS>(pX[1], row, c, vb);
            store_split<SP, ROWS>(pX[2], row, c, vs);
        }
        __syncthreads();

        // y = relu(Wc x + bc), 384 -> 384 in three 128-column groups
#pragma unroll 1
        for (int g = 0; g < 3; ++g) {
            f4 acc[NRB][1];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = g == 0 ? cb[0] : g == 1 ? cb[1] : cb[2];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                split_raw();
                const int un = 3 * g + kt + 1;                  // next unit; 9..11 only exist for wavefronts 0..3
                if (un < 9 || wv < 4) issue(un);
                mma_tile_split<SP, 4, 1, NRB, ROWS, 256, 4, 0, true, 3>(pX[kt], wf, acc, lane);
            }
            bool ybad = false;   // this kernel's OWN f16x2 operands: an activation (or a weight, via a NaN accumulator) beyond the fp16
#pragma unroll           // range would come out of the ReLUs below as a finite, wrong ddG
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ybad = ybad || tm_f16_range_computed(acc[rb][0][k]);
                store_split<SP, ROWS>(pY[g], 16 * rb + m, c4, relu4(acc[rb][0]));
            }
            if (a.status && ybad) atomicOr(a.status, TMPNN_STATUS_RANGE);
        }
        __syncthreads();

        if (wv < 4) {   // 384 -> 64, relu; wavefront w owns columns 16w..16w+15 -> tF0[:, 0:64] (x planes are dead)
            f4 acc[NRB][1];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b1v;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                split_raw();
                if (kt < 2) issue(10 + kt);
                mma_tile_split<SP, 4, 1, NRB, ROWS, 256, 4, 0, true, 3>(pY[kt], wf, acc, lane);
            }
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) st4(tF0 + chunk_off(16 * rb + m, c4), relu4(acc[rb][0]));
        }
        if (tile + (int)tm_nblk() < n_tiles) issue(0);          // unit 0 of this workgroup's next tile
        __syncthreads();
        if (wv < 2) {   // 64 -> 32, relu -> tF1[:, 0:32]
            f4 acc[NRB][1];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b2v;
            mma_tile<4, 1, 128, NRB>(tF0, w32, acc, lane);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) st4(tF1 + chunk_off(16 * rb + m, c4), relu4(acc[rb][0]));
        }
        __syncthreads();
        if (wv < 2) {   // 32 -> 21 (rows 21..31 of the weight read as zero) -> z in tF2[:, 0:32]
            f4 acc[NRB][1];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b3v;
            mma_tile<2, 1, 128, NRB>(tF1, w8, acc, lane);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) st4(tF2 + chunk_off(16 * rb + m, c4), acc[rb][0]);
        }
        __syncthreads();
        for (int e = tid; e < rows * TMPNN_VOCAB; e += 512) {
            const int row = e / TMPNN_VOCAB, aa = e - row * TMPNN_VOCAB;
            const float z = tF2[chunk_off(row, aa >> 2) + (aa & 3)];
            const int wt = s_S[row];
            const float zw = tF2[chunk_off(row, wt >> 2) + (wt & 3)];
            const float dd = (dw * z + db) - (dw * zw + db);   // :110-116
            a.ddg[(size_t)(r0 + row) * TMPNN_VOCAB + aa] = dd;
            if (a.status && tm_nonfinite(dd)) atomicOr(a.status, TMPNN_STATUS_RANGE);
            if (a.z_opt) a.z_opt[(size_t)(r0 + row) * TMPNN_VOCAB + aa] = z;
        }
        __syncthreads();
    }
}

