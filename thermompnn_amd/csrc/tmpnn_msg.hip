// Message pass of an encoder / decoder layer (protein_mpnn_utils.py:816-823, 859-866), split-precision forms: f16x2 = msg8_rp_kernel,
// bf16x3 = msg8_split_kernel.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "tmpnn_split.h"
#include "tmpnn_internal.h"

// ------------------------------------------------------------------------------------------------
// message kernels, split-precision form (8 wavefronts, 1 workgroup per CU, next tile prefetched through registers).
// Same arithmetic as msg_kernel (tmpnn_layers.hip): Ssum_i = sum_k ma_ik gelu(W2 gelu(pre_ik) + b2).
// ------------------------------------------------------------------------------------------------
template <typename SP, bool DEC>
__global__ __launch_bounds__(512, 2) void msg8_split_kernel(MsgArgsB a) {
    constexpr int TILEB = SP::NP * SPLIT_PLANE_BYTES;
    __shared__ __attribute__((aligned(16))) char tE[TILEB];
    __shared__ __attribute__((aligned(16))) char tA[TILEB];
    __shared__ __attribute__((aligned(16))) float tS[TM_TILE * TM_H];
    __shared__ __attribute__((aligned(16))) float tStage[TM_TILE * TM_H];   // next residue's fp32 tile, landed by LDS-DMA
    __shared__ float s_part[3][TM_H];
    __shared__ int s_idx[2][TM_TILE];
    __shared__ float s_ma[2][TM_TILE];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;

    WFragS<SP> w1[1][4], w2[1][4];
    load_wfrag_split<SP, 4>(a.W1e, a.ld1, 16 * wv, 0, TM_H, w1[0], lane);
    load_wfrag_split<SP, 4>(a.W2, TM_H, 16 * wv, 0, TM_H, w2[0], lane);
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const f4 bias2 = ld4(a.b2 + ncol);

    auto stage_async = [&](const float *src) {        // linear LDS-DMA of one fp32 tile: 24 x 1 KB, three per wavefront
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int blk = 3 * wv + k;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + blk * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(tStage + blk * 256), 16, 0, 0);
        }
    };
    auto split_stage = [&]() {
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = it * 512 + tid;
            store_split<SP>(tE, idx >> 5, idx & 31, ld4(tStage + idx * 4));
        }
    };
    auto stage_idx = [&](int ii, int buf) {           // neighbour list + attention mask of residue ii -> LDS
        if (tid < TM_TILE) {
            const int j = a.E_idx[(size_t)ii * TM_KS + tid];
            s_idx[buf][tid] = j;
            s_ma[buf][tid] = j < 0 ? 0.f : (DEC ? 1.f : a.mask[ii] * a.mask[j]);
        }
    };
    f4 g0, gj[3];                                      // node terms of the tile about to be processed
    auto gather = [&](int ii, int buf) {
        g0 = ld4(a.P + (size_t)ii * 256 + ncol);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int j0 = s_idx[buf][16 * rb + m];
            const int j = j0 < 0 ? ii : j0;
            gj[rb] = ld4(a.P + (size_t)j * 256 + 128 + ncol);
        }
    };

    const TileRange tr = xcd_tile_range(a.T);
    int i = tr.begin;
    int cur = 0;
    if (i < tr.end) {
        stage_idx(i, 0);
        stage_async(a.hE + (size_t)i * TM_KS * TM_H);
        __syncthreads();
        split_stage();
        gather(i, 0);
        __syncthreads();
    }
    for (; i < tr.end; i += tr.step) {
        const int inext = i + tr.step;
        const bool has_next = inext < tr.end;
        const float mi = a.mask[i];
        if (has_next) {
            stage_async(a.hE + (size_t)inext * TM_KS * TM_H);
            stage_idx(inext, cur ^ 1);
        }
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = DEC ? gj[rb] : g0 + gj[rb];
        mma_tile_split<SP, 4, 1>(tE, w1, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            f4 v = acc[rb][0];
            if (DEC) v = g0 + mi * v;
            store_split<SP>(tA, 16 * rb + m, c4, gelu4(v));
        }
        __syncthreads();                                         // tE consumed; tA, tStage, s_idx/s_ma[next] complete

        if (has_next) {
            split_stage();
            gather(inext, cur ^ 1);
        }
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = bias2;
        mma_tile_split<SP, 4, 1>(tA, w2, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const float ma = s_ma[cur][16 * rb + m];
            f4 v = gelu4(acc[rb][0]) * ma;
            if (ma == 0.f) v = f4{0.f, 0.f, 0.f, 0.f};
            st4(tS + chunk_off(16 * rb + m, c4), v);
        }
        if (wv == 2) {                                           // neighbour count of this tile (read before s_ma[cur] is recycled):
            float c = lane < TM_TILE ? s_ma[cur][lane] : 0.f;    // one wavefront-wide DPP sum (a serial 48-term loop in one lane
#define TM_DPP_ADD(ctrl, row_mask, bc)                                                                  \
            c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), ctrl, row_mask, 0xf, bc));
            TM_DPP_ADD(0x111, 0xf, true)                         // held the other seven wavefronts at the barrier for ~350 cycles)
            TM_DPP_ADD(0x112, 0xf, true)
            TM_DPP_ADD(0x114, 0xf, true)
            TM_DPP_ADD(0x118, 0xf, true)                         // lane 15 of every row: the row's sum
            TM_DPP_ADD(0x142, 0xa, false)                        // row_bcast:15 into rows 1 and 3
            TM_DPP_ADD(0x143, 0xc, false)                        // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef TM_DPP_ADD
            if (lane == 63) a.cnt[i] = c;
        }
        __syncthreads();
        {   // per-node aggregation: column sums over 4 row groups of 12, combined in a fixed order
            const int n = tid & 127, grp = tid >> 7;
            float s = 0.f;
#pragma unroll
            for (int r = 12 * grp; r < 12 * grp + 12; ++r) s += tS[chunk_off(r, n >> 2) + (n & 3)];
            if (grp) s_part[grp - 1][n] = s;
            __syncthreads();
            if (!grp) a.Ssum[(size_t)i * TM_H + n] = ((s + s_part[0][n]) + s_part[1][n]) + s_part[2][n];
        }
        cur ^= 1;
        // no barrier here: the next iteration writes tA only after its own GEMM1 (behind which every wavefront has
        // passed the barrier above), tS / s_part only after two more barriers, and s_idx/s_ma[cur^1] = the buffers
        // of the iteration before this one.
    }
}

// Register-prefetch form of the message kernel (f16x2): the next residue's fp32 tile is loaded in the accumulator
// layout at the top of the iteration and split into the e planes once GEMM 1 has consumed the current ones.
// OFF32: the node-projection table is smaller than 4 GB (T < 2^22 rows), so a gathered row is addressed as the uniform table
// pointer + a 32-bit per-lane byte offset (one VALU op per gather instead of a 64-bit shift + add chain); every other global access
// of the loop is a wave-uniform base + a per-thread offset computed once, whatever T is.
template <typename SP, bool DEC, bool PROF = false, bool OFF32 = false>
__global__ __launch_bounds__(512, 2) void msg8_rp_kernel(MsgArgsB a, unsigned long long *prof = nullptr) {
    unsigned long long t_last = 0;
    auto mark = [&](int k) {           // TMPNN_MSG_PROF=1: phase timing of thread 0 of workgroup 0
        if (PROF && tm_bid() == 0 && tm_tid() == TM_PROF_TID) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (k >= 0) prof[k] += t - t_last;
            t_last = t;
        }
    };
    constexpr int TILEB = SP::NP * SPLIT_PLANE_BYTES;
    __shared__ __attribute__((aligned(16))) char tE[TILEB];
    __shared__ __attribute__((aligned(16))) char tA[TILEB];
    __shared__ int s_idx[2][TM_TILE];
    __shared__ float s_ma[2][TM_TILE];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;

    WFragS<SP> w1[1][4], w2[1][4];
    load_wfrag_auto<SP>(a.img1, a.W1e, a.ld1, wv, lane, w1[0]);
    load_wfrag_auto<SP>(a.img2, a.W2, TM_H, wv, lane, w2[0]);
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const f4 bias2 = ld4(a.b2 + ncol);

    auto stage_idx = [&](int ii, int buf) {           // neighbour list + attention mask of residue ii -> LDS
        if (tid < TM_TILE) {
            const int j = (a.E_idx + (size_t)__builtin_amdgcn_readfirstlane(ii) * TM_KS)[(unsigned)tid];
            s_idx[buf][tid] = j;
            s_ma[buf][tid] = j < 0 ? 0.f : (DEC ? 1.f : a.mask[ii] * a.mask[j]);
        }
    };
    f4 g0, gj[3], e_nxt[3];
    const unsigned ucol = (unsigned)ncol;
    auto gather = [&](int ii, int buf) {
        g0 = ld4(a.P + (size_t)__builtin_amdgcn_readfirstlane(ii) * 256 + ucol);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int j0 = s_idx[buf][16 * rb + m];
            const int j = j0 < 0 ? ii : j0;
            if constexpr (OFF32) gj[rb] = ld4(a.P + ((unsigned)j * 256u + (128u + ucol)));
            else gj[rb] = ld4(a.P + (size_t)j * 256 + 128 + ncol);
        }
    };
    // row layout: one half-wavefront per 512-byte row, fully coalesced (the message kernels never need the tile in
    // the accumulator layout)
    const int prow = 6 * wv + (lane >> 5), pc = lane & 31;
    const unsigned eoff = (unsigned)(prow * TM_H + 4 * pc);         // this thread's offset inside any e tile
    auto fetch_into = [&](f4 (&dst)[3], int ii) {
        const float *src = a.hE + (size_t)__builtin_amdgcn_readfirstlane(ii) * (TM_KS * TM_H);      // wave-uniform: scalar base + lane offset
#pragma unroll
        for (int it = 0; it < 3; ++it) dst[it] = ld4(src + (eoff + 2 * it * TM_H));
    };
    auto fetch_tile = [&](int ii) { fetch_into(e_nxt, ii); };
    auto split_tile = [&]() {
#pragma unroll
        for (int it = 0; it < 3; ++it) store_split<SP>(tE, prow + 2 * it, pc, e_nxt[it]);
    };

    const TileRange tr = xcd_tile_range(a.T);
    int i = tr.begin;
    int cur = 0;
    if (i < tr.end) {
        stage_idx(i, 0);
        fetch_tile(i);
        __syncthreads();
        split_tile();
        gather(i, 0);
        fetch_tile(i + tr.step < tr.end ? i + tr.step : i);     // e_nxt always holds the tile AFTER the one in the planes
        __syncthreads();
    }
    // mask of the residue in the planes: requested one iteration before it is used (gfx9 waits for loads in order — fetched at the
    // top of its own iteration it cost a vmcnt(0) right behind GEMM 1)
    float mi = i < tr.end ? a.mask[i] : 0.f;
    mark(-1);
    for (; i < tr.end; i += tr.step) {
        const int inext = i + tr.step;
        const int ipf = inext < tr.end ? inext : i;             // the last iteration prefetches its own tile again
        // neighbour list of the next residue: loaded first, its dependent mask gather REQUESTED behind GEMM 1 and USED behind the
        // epilogue, both published to LDS just in front of the barrier — no wavefront sits on a global-load latency. (Round 5: with
        // the product mask[ipf] * mask[nidx] formed inside the `tid < 48` branch hipcc waited for the gather right where it was
        // issued: wavefront 0 sat out a whole L2 round trip per tile in front of its GELU, the other seven at the barrier.)
        int nidx = -1;
        if (tid < TM_TILE) nidx = (a.E_idx + (size_t)__builtin_amdgcn_readfirstlane(ipf) * TM_KS)[(unsigned)tid];
        const float mi_nxt = (a.mask + __builtin_amdgcn_readfirstlane(ipf))[0];
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = DEC ? gj[rb] : g0 + gj[rb];
        mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_MSG_PF>(tE, w1, acc, lane);
        mark(0);
        float mk_j = 1.f;
        if (!DEC && tid < TM_TILE) mk_j = a.mask[(unsigned)(nidx >= 0 ? nidx : ipf)];       // requested only; first use below
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            f4 v = acc[rb][0];
            if (DEC) v = g0 + mi * v;
            store_split<SP>(tA, 16 * rb + m, c4, gelu4(v));
        }
        if (tid < TM_TILE) {
            s_idx[cur ^ 1][tid] = nidx;
            s_ma[cur ^ 1][tid] = nidx >= 0 ? (DEC ? 1.f : mi_nxt * mk_j) : 0.f;
        }
        mark(1);
        __syncthreads();                                         // tE consumed; tA, s_idx/s_ma[next] complete
        mark(2);

        split_tile();
        // Order matters (gfx9 retires loads in order): the node-term gathers of the NEXT tile first, then the request for the
        // tile after the next. The gathers are waited for at the end of this iteration; were they younger than the tile
        // loads, that wait would also force the tile loads home after one GEMM phase instead of one full iteration (an HBM
        // round trip under load is longer than a phase: ablation showed only 0.02 of the 0.09 ms of e-tile streaming hidden).
        gather(ipf, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);                       // (hipcc hoisted the tile request above the gathers: the wait for the
                                                                 //  gathers at the end of the iteration then drained it too — vmcnt(0))
#if !TM_ABL_NOLOAD
        fetch_tile(ipf + tr.step < tr.end ? ipf + tr.step : ipf);
#endif
        mark(3);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = bias2;
        mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_MSG_PF>(tA, w2, acc, lane);
        mark(4);
        f4 tot = f4{0.f, 0.f, 0.f, 0.f};                         // masked sum over the K neighbours, in registers
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {                         // tot += ma * gelu(...): one fma per value. (ma is 0 or 1, so the
            const float ma = s_ma[cur][16 * rb + m];            // product is exact and this IS the reference's mask_attend * h_message,
            const f4 g = gelu4(acc[rb][0]);                      // :821-823 — a separate "select 0 where ma == 0" cost 5 more VALU per row block)
            tot = f4{__builtin_fmaf(g.x, ma, tot.x), __builtin_fmaf(g.y, ma, tot.y), __builtin_fmaf(g.z, ma, tot.z), __builtin_fmaf(g.w, ma, tot.w)};
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {                             // inclusive scan over the 16 rows of the lane group (DPP row_shr,
            float x = tot[c];                                    // zero fill): lane m = 15 ends up with the column sum
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
            tot[c] = x;
        }
        {                                                        // take the gathers' vmcnt wait before any store is issued (see touch())
            touch(g0);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) touch(gj[rb]);
        }
        if (m == 15) st4(a.Ssum + (size_t)__builtin_amdgcn_readfirstlane(i) * TM_H + ucol, tot);
        if (wv == 2) {                                           // neighbour count of this tile (read before s_ma[cur] is recycled):
            float c = lane < TM_TILE ? s_ma[cur][lane] : 0.f;    // one wavefront-wide DPP sum (a serial 48-term loop in one lane
#define TM_DPP_ADD(ctrl, row_mask, bc)                                                                  \
            c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), ctrl, row_mask, 0xf, bc));
            TM_DPP_ADD(0x111, 0xf, true)                         // held the other seven wavefronts at the barrier for ~350 cycles)
            TM_DPP_ADD(0x112, 0xf, true)
            TM_DPP_ADD(0x114, 0xf, true)
            TM_DPP_ADD(0x118, 0xf, true)                         // lane 15 of every row: the row's sum
            TM_DPP_ADD(0x142, 0xa, false)                        // row_bcast:15 into rows 1 and 3
            TM_DPP_ADD(0x143, 0xc, false)                        // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef TM_DPP_ADD
            if (lane == 63) a.cnt[i] = c;
        }
        mark(5);
        mark(6);
        cur ^= 1;
        mi = mi_nxt;
        __syncthreads();                                         // tA consumed (the next GEMM-1 epilogue rewrites it), tE complete
        mark(7);
    }
}

int launch_msg_split(int mode, bool dec, const float *W1e, int ld1, const float *W2, const float *b2, const float *P,
                     const float *hE, const int32_t *E_idx, const float *mask, int64_t T, float *Ssum, float *cnt,
                     hipStream_t st) {
    const bool h2 = mode == TM_MM_F16X2;
    MsgArgsB a{W1e, ld1, W2, b2, P, hE, E_idx, mask, Ssum, cnt, (int)T, h2 ? tm_find_wimg(W1e) : nullptr, h2 ? tm_find_wimg(W2) : nullptr};
    const int64_t cap = tm_num_cus();
    const int grid = (int)(T < cap ? T : cap);
    if (mode == TM_MM_BF16X3) {                      // staged through LDS (the exact three-plane tiles leave no VGPRs for a register prefetch)
        if (dec) msg8_split_kernel<SplitBF3, true><<<grid, 512, 0, st>>>(a);
        else msg8_split_kernel<SplitBF3, false><<<grid, 512, 0, st>>>(a);
    } else {
#ifdef TMPNN_DEBUG_BUILD
        static const bool prof = TM_DBG_FLAG("TMPNN_MSG_PROF", false);
#else
        constexpr bool prof = false;
#endif
        if (prof && dec) {                           // debug build: phase timing of workgroup 0 (synchronises!)
#ifdef TMPNN_DEBUG_BUILD
            static unsigned long long *d_prof = nullptr;
            if (!d_prof) (void)hipMalloc(&d_prof, 16 * sizeof(unsigned long long));
            (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), st);
            msg8_rp_kernel<SplitH2, true, true, false><<<grid, 512, 0, st>>>(a, d_prof);
            unsigned long long h[16];
            (void)hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "dec_msg phases (cycles, wg 0): fetch+gemm1 %llu gelu+split %llu bar %llu split_tile+gather %llu gemm2 %llu gelu+mask %llu bar %llu ksum+store %llu\n",
                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
#endif
        } else if (T < ((int64_t)1 << 22)) {           // projection table < 4 GB: 32-bit gather offsets
            if (dec) msg8_rp_kernel<SplitH2, true, false, true><<<grid, 512, 0, st>>>(a);
            else msg8_rp_kernel<SplitH2, false, false, true><<<grid, 512, 0, st>>>(a);
        } else if (dec) msg8_rp_kernel<SplitH2, true><<<grid, 512, 0, st>>>(a);
        else msg8_rp_kernel<SplitH2, false><<<grid, 512, 0, st>>>(a);
    }
    return tm_check_launch(dec ? "dec_msg_split" : "enc_msg_split");
}
