// Split-precision forms of the per-edge kernels (f16x2 three-term / bf16x3 six-term). See tmpnn_split.h for the arithmetic.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "tmpnn_split.h"

#ifndef TM_SETPRIO
#define TM_SETPRIO 0   // 1: s_setprio 1 for wavefronts 4-7 of the per-edge kernels (VALU arbitration is by age; the guide reports -0.8..-1.5 % for an attention loop). Measured here with a provably wave-uniform condition: +2.5 % message kernels, +2 % edge update - off
#endif
#ifndef TM_PROF_TID
#define TM_PROF_TID 0   // thread of workgroup 0 whose cycle counter the TMPNN_*_PROF phase timers read (448 = wavefront 7, lowest issue priority)
#endif
#ifndef TM_EDGE_UNROLL2
#define TM_EDGE_UNROLL2 0   // 1: the edge update's tile loop unrolled by two with the e-tile register sets swapping roles (no loop-carried copy:
                            // 422 instead of 437 VALU per tile, 234 VGPRs, no spill). Measured in one call, three alternations: 0.2948 vs 0.2932 ms — nil; off
#endif
#ifndef TM_EDGE_Y_ALIAS
#define TM_EDGE_Y_ALIAS 1   // 0: a third plane tile for GEMM 2's output (77.5 KB of LDS, offsets above 64 KB): the rounds 1-3 form, A/B only
#endif
#ifndef TM_EDGE_PF
#define TM_EDGE_PF 2     // the same for the edge-update kernel's three 12-step GEMMs: 0.339 ms at 0, 0.329 at 1, 0.323 at 2, 0.326 at 3-4
#endif
#ifndef TM_NODE_PF
#define TM_NODE_PF 3
#endif
#ifndef TM_NODE_DEEP_D
#define TM_NODE_DEEP_D 3   // fragment images in flight ahead of the GEMM unit being computed (node_update8_deep_kernel)
#endif
#ifndef TM_NODE_DEEP_PF
#define TM_NODE_DEEP_PF 2  // its B-fragment prefetch distance (a 16-row GEMM has 4 steps; 3 would hold all four at once: 8 more VGPRs)
#endif
#ifndef TM_MSG_TOUCH
#define TM_MSG_TOUCH 1
#endif
// timing-only ablations of the per-edge kernels (wrong results; tools/ablate_build.sh): each removes ONE ingredient so that
// its true cost in the pipeline shows as a time difference — TM_ABL_NOGELU (tmpnn_common.h), TM_ABL_NOSPLIT / TM_ABL_NOMFMA
// (tmpnn_split.h), TM_ABL_NOLN (edge update: no LayerNorm statistics)
#ifndef TM_ABL_NOLN
#define TM_ABL_NOLN 0
#endif
#ifndef TM_ABL_NOLOAD
#define TM_ABL_NOLOAD 0      // timing-only: the per-edge kernels never fetch the NEXT e tile (they keep re-using the first one)
#endif
#ifndef TM_MSG_PFD
#define TM_MSG_PFD 1         // message kernel: e tiles requested this many tiles ahead of the one in the planes (registers: 12 VGPRs each)
#endif
#ifndef TM_MSG_PF
#define TM_MSG_PF 3      // B-fragment prefetch distance (steps) of the 8-wavefront message kernel GEMMs, see mma_tile_split
#endif
#include "tmpnn_internal.h"

// ------------------------------------------------------------------------------------------------
// GEMM probe (tools/gemm_probe.py): Y[t] = X[t] W^T per 48 x 128 tile, 8 wavefronts.
// MODE 0 = fp32 MFMA, 1 = bf16x3 six-term, 2 = f16x2 three-term
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm_probe_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                            float *__restrict__ Y, int T, int reps) {
    __shared__ __attribute__((aligned(16))) float tF[TM_TILE * TM_H];
    using SP = typename std::conditional<MODE == 2, SplitH2, SplitBF3>::type;
    __shared__ __attribute__((aligned(16))) char tP[3 * SPLIT_PLANE_BYTES];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    float wf[1][32];
    WFragS<SP> w3[1][4];
    if (MODE == 0) load_wfrag<8>(W, TM_H, 16 * wv, 0, TM_H, wf[0], lane);
    else load_wfrag_split<SP, 4>(W, TM_H, 16 * wv, 0, TM_H, w3[0], lane);
    for (int i = tm_bid(); i < T; i += tm_nblk()) {
        const float *src = X + (size_t)i * TM_TILE * TM_H;
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = it * 512 + tid, row = idx >> 5, c = idx & 31;
            const f4 v = ld4(src + (size_t)idx * 4);
            if (MODE == 0) st4(tF + chunk_off(row, c), v);
            else store_split<SP>(tP, row, c, v);
        }
        __syncthreads();
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = f4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < reps; ++r) {
            if (MODE == 0) mma_tile<8, 1>(tF, wf, acc, lane);
            else mma_tile_split<SP, 4, 1>(tP, w3, acc, lane);
        }
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
            st4(Y + ((size_t)i * TM_TILE + 16 * rb + m) * TM_H + 16 * wv + 4 * q, acc[rb][0]);
        __syncthreads();
    }
}

// Device self-test of the two things a different compiler / flag set could silently break:
//  (1) the f16x2 kernels of THIS translation unit detect an fp16 overflow because their GELU propagates NaN — its clamps must
//      have lowered to gfx950's v_minimum3 / v_maximum3 (build.py FILE_FLAGS: no -fno-honor-nans here). gelu(NaN) must be NaN and
//      gelu(+inf) non-finite. (gelu(-inf) = -t 2^P(t) + max(-inf, 0) = -4e-8: finite and intended — an overflow reaches GELU as
//      NaN, never as a lone inf: h = +-inf and l = -+inf meet in one accumulator.)
//  (2) tm_nblk() / tm_bdim() read gridDim / blockDim from fixed offsets of the code-object-v5 implicit-argument block
//      (build.py pins -mcode-object-version=5); every persistent tile loop strides by them.
// Inputs arrive as kernel arguments so nothing folds at compile time. ORs TMPNN_STATUS_SELFTEST into *status on failure.
__global__ void selftest_kernel(unsigned nan_bits, unsigned inf_bits, int32_t *status) {
    bool bad = tm_nblk() != (int)gridDim.x || tm_bdim() != (int)blockDim.x;
    const f2 g = gelu2(f2{__uint_as_float(nan_bits), __uint_as_float(inf_bits)});
    float gx = g.x, gy = g.y;
    asm volatile("" : "+v"(gx), "+v"(gy));
    const unsigned bx = __float_as_uint(gx), by = __float_as_uint(gy);
    bad = bad || !((bx & 0x7f800000u) == 0x7f800000u && (bx & 0x007fffffu) != 0u);     // NaN in -> NaN out
    bad = bad || (by & 0x7f800000u) != 0x7f800000u;                                      // +inf in -> non-finite out
    if (bad && tm_tid() == 0) atomicOr(status, TMPNN_STATUS_SELFTEST);
}
int launch_selftest(int32_t *status, hipStream_t st) {
    selftest_kernel<<<3, 128, 0, st>>>(0x7fc00000u, 0x7f800000u, status);
    return tm_check_launch("selftest");
}

int launch_gemm_probe(int mode, const float *X, const float *W, float *Y, int64_t T, int reps, hipStream_t st) {
    const int64_t cap = tm_num_cus();
    const int grid = (int)(T < cap ? T : cap);
    if (mode == 0) gemm_probe_kernel<0><<<grid, 512, 0, st>>>(X, W, Y, (int)T, reps);
    else if (mode == 1) gemm_probe_kernel<1><<<grid, 512, 0, st>>>(X, W, Y, (int)T, reps);
    else gemm_probe_kernel<2><<<grid, 512, 0, st>>>(X, W, Y, (int)T, reps);
    return tm_check_launch("gemm_probe");
}


// ------------------------------------------------------------------------------------------------
// enc_edge, split-precision form (8 wavefronts, 1 workgroup per CU): same pipeline as enc_edge8_kernel
// (tmpnn_layers.hip) with the three 128x128 GEMMs on the 16-bit matrix cores. GEMM inputs live in LDS as plane tiles;
// the LayerNorm input is an fp32 tile aliased on the x planes. The next residue's fp32 tile lands in an LDS staging
// buffer by LDS-DMA under GEMM 1 and is split into the e planes during the LayerNorm/store phase. Residual: bf16x3
// re-joins the e planes (exact); f16x2 keeps the fp32 tile (two staging buffers, alternating).
// ------------------------------------------------------------------------------------------------
// Weight fragment of wavefront wv from a pre-built image (WImg, tmpnn_internal.h): 8 coalesced 16-byte loads instead of the
// 16-row fp32 gathers + on-the-fly split of load_wfrag_split. img == nullptr -> the gather path.
template <typename SP>
__device__ __forceinline__ void load_wfrag_auto(const char *img, const float *__restrict__ W, int ld, int wv, int lane,
                                                WFragS<SP> (&wf)[4]) {
    if (img != nullptr && SP::NP == 2) {
        const char *p = img + (size_t)wv * 8192 + lane * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            wf[c].p[0] = *reinterpret_cast<const u4 *>(p + 2048 * c);
            wf[c].p[1] = *reinterpret_cast<const u4 *>(p + 2048 * c + 1024);
        }
    } else {
        load_wfrag_split<SP, 4>(W, ld, 16 * wv, 0, TM_H, wf, lane);
    }
}

struct EdgeArgsB {
    const float *W11e, *W12, *b12, *W13, *b13, *g3, *be3, *P;
    float *hE;
    const int32_t *E_idx;
    int T;
    const char *img11, *img12, *img13;      // fragment images of the three weights (f16x2 only) or null
};

template <typename SP>
__global__ __launch_bounds__(512, 2) void enc_edge8_split_kernel(EdgeArgsB a) {
    constexpr int TILEB = SP::NP * SPLIT_PLANE_BYTES;
    constexpr int NST = SP::EXACT ? 1 : 2;                               // fp32 staging buffers
    static_assert(TILEB >= TM_TILE * TM_H * 4, "the fp32 LayerNorm tile is aliased on the x planes");
    __shared__ __attribute__((aligned(16))) char tE[TILEB];
    __shared__ __attribute__((aligned(16))) char tX[TILEB];              // x planes; later the fp32 LayerNorm input
    __shared__ __attribute__((aligned(16))) char tY[TILEB];
    __shared__ __attribute__((aligned(16))) float tStageB[NST][TM_TILE * TM_H]; // fp32 tiles landed by LDS-DMA
    __shared__ __attribute__((aligned(16))) float s_stat[TM_TILE][TM_STAT8_LD];
    __shared__ int s_idx[2][TM_TILE];
    float *tO = reinterpret_cast<float *>(tX);
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;

    WFragS<SP> w11[1][4], w12[1][4], w13[1][4];
    load_wfrag_split<SP, 4>(a.W11e, 384, 16 * wv, 0, TM_H, w11[0], lane);
    load_wfrag_split<SP, 4>(a.W12, TM_H, 16 * wv, 0, TM_H, w12[0], lane);
    load_wfrag_split<SP, 4>(a.W13, TM_H, 16 * wv, 0, TM_H, w13[0], lane);
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const int c32 = lane & 31;

    // linear (unswizzled) LDS-DMA of one fp32 tile: 24 wave-instructions of 1 KB, three per wavefront
    auto stage_async = [&](const float *src, float *tStage) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int blk = 3 * wv + k;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + blk * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(tStage + blk * 256), 16, 0, 0);
        }
    };
    auto split_stage = [&](const float *tStage) {      // tStage (fp32, linear) -> e planes
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = it * 512 + tid;
            store_split<SP>(tE, idx >> 5, idx & 31, ld4(tStage + idx * 4));
        }
    };

    const TileRange tr = xcd_tile_range(a.T);
    int i = tr.begin;
    int cur = 0, sb = 0;                               // s_idx buffer / staging buffer of the current tile
    f4 gai, gcj[3];
    if (i < tr.end) {
        if (tid < TM_TILE) s_idx[0][tid] = a.E_idx[(size_t)i * TM_KS + tid];
        stage_async(a.hE + (size_t)i * TM_KS * TM_H, tStageB[0]);
        __syncthreads();
        split_stage(tStageB[0]);
        gai = ld4(a.P + (size_t)i * 256 + ncol);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int j = s_idx[0][16 * rb + m];
            gcj[rb] = ld4(a.P + (size_t)(j < 0 ? i : j) * 256 + 128 + ncol);
        }
        __syncthreads();
    }
    for (; i < tr.end; i += tr.step) {
        float *tile_g = a.hE + (size_t)i * TM_KS * TM_H;
        const int inext = i + tr.step;
        const bool has_next = inext < tr.end;
        int nidx = -1;
        const int sn = NST == 2 ? sb ^ 1 : 0;
        if (has_next) {
            stage_async(a.hE + (size_t)inext * TM_KS * TM_H, tStageB[sn]);
            if (tid < TM_TILE) nidx = a.E_idx[(size_t)inext * TM_KS + tid];
        }
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = gai + gcj[rb];
        mma_tile_split<SP, 4, 1>(tE, w11, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            store_split<SP>(tX, 16 * rb + m, c4, gelu4(acc[rb][0]));
            __builtin_amdgcn_sched_barrier(0);      // one row block at a time: keeps the GELU temporaries out of the weight VGPRs
        }
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
        {
            const f4 b12 = ld4(a.b12 + ncol);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b12;
        }
        mma_tile_split<SP, 4, 1>(tX, w12, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            store_split<SP>(tY, 16 * rb + m, c4, gelu4(acc[rb][0]));
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();

        {
            const f4 b13 = ld4(a.b13 + ncol);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b13;
        }
        mma_tile_split<SP, 4, 1>(tY, w13, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int row = 16 * rb + m;
            const f4 e = SP::EXACT ? load_joined<SP>(tE, row, c4)            // residual: exact re-join of the e planes
                                   : ld4(tStageB[sb] + row * TM_H + 4 * c4); //           or the fp32 tile itself
            const f4 v = e + acc[rb][0];
            st4(tO + chunk_off(16 * rb + m, c4), v);
            row_stats_partial1b(v, &s_stat[16 * rb + m][2 * wv], q);
        }
        __syncthreads();                                                     // tE free, tO + stats complete

        if (has_next) split_stage(tStageB[sn]);
        {
            const f4 g4 = ld4(a.g3 + 4 * c32), be4 = ld4(a.be3 + 4 * c32);
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int row = 6 * wv + 2 * it + (lane >> 5);
                float mean, rstd;
                row_stats_finish8b(&s_stat[row][0], mean, rstd);
                const f4 y = (ld4(tO + chunk_off(row, c32)) - mean) * rstd * g4 + be4;
                if (s_idx[cur][row] >= 0) st4(tile_g + (size_t)row * TM_H + 4 * c32, y);
            }
        }
        cur ^= 1;
        sb = sn;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// enc_edge, register-prefetch form (used for f16x2, which leaves the VGPRs for it): the next residue's fp32 tile is
// loaded straight into the accumulator layout (row 16 rb + m, columns 16 wv + 4 q: one 16-byte load per row block)
// at the top of the iteration, split into the e planes after GEMM 3 and kept in registers as the fp32 residual of the
// next iteration. No LDS staging, no LDS-DMA (whose conservative vmcnt(0) waits serialised the store phase), biases and
// LayerNorm parameters live in registers, every global access of the loop is unconditional.
// ------------------------------------------------------------------------------------------------
// OFF32: see msg8_rp_kernel (32-bit gather offsets when the projection table is smaller than 4 GB).
template <typename SP, bool PROF = false, bool OFF32 = false>
__global__ __launch_bounds__(512, 2) void enc_edge8_rp_kernel(EdgeArgsB a, unsigned long long *prof = nullptr) {
    unsigned long long t_last = 0;
    auto mark = [&](int k) {           // TMPNN_EDGE_PROF=1: phase timing of thread 0 of workgroup 0
        if (PROF && tm_bid() == 0 && tm_tid() == TM_PROF_TID) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (k >= 0) prof[k] += t - t_last;
            t_last = t;
        }
    };
    constexpr int TILEB = SP::NP * SPLIT_PLANE_BYTES;
    static_assert(TILEB >= TM_TILE * TM_H * 4, "the fp32 LayerNorm tile is aliased on the x planes");
    __shared__ __attribute__((aligned(16))) char tE[TILEB];
    __shared__ __attribute__((aligned(16))) char tX[TILEB];              // x planes; later the fp32 LayerNorm input
    // GEMM 2's output planes live where the e planes were: GEMM 1 was their last reader (every wavefront is past the barrier behind
    // it), the next tile's e planes are written only behind the barrier that follows GEMM 3. Two plane tiles instead of three:
    // 53 KB of LDS, every LDS offset below 64 KB (an offset above costs an address VGPR + a v_or each: 10 VALU per tile).
#if TM_EDGE_Y_ALIAS
    char *const tY = tE;
#else
    __shared__ __attribute__((aligned(16))) char tY[TILEB];
#endif
    __shared__ __attribute__((aligned(16))) float s_stat[TM_TILE][TM_STAT_LD];
    __shared__ int s_idx[2][TM_TILE];
    float *tO = reinterpret_cast<float *>(tX);
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;

    WFragS<SP> w11[1][4], w12[1][4], w13[1][4];
    load_wfrag_auto<SP>(a.img11, a.W11e, 384, wv, lane, w11[0]);
    load_wfrag_auto<SP>(a.img12, a.W12, TM_H, wv, lane, w12[0]);
    load_wfrag_auto<SP>(a.img13, a.W13, TM_H, wv, lane, w13[0]);
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const int c32 = lane & 31;
    const unsigned ucol = (unsigned)ncol;
    const unsigned eoff = (unsigned)(m * TM_H + ncol);                    // this thread's offset inside an e tile (accumulator layout, row block 0)
    const unsigned soff = (unsigned)((6 * wv + (lane >> 5)) * TM_H + 4 * c32);   // ... in the row layout of the LayerNorm / store phase
    auto prow_of = [&](int j, int self) -> const float * {                // &P[j][128 + ncol] (j < 0: the residue's own row)
        const int jj = j < 0 ? self : j;
        if constexpr (OFF32) return a.P + ((unsigned)jj * 256u + (128u + ucol));
        else return a.P + (size_t)jj * 256 + 128 + ncol;
    };
    const f4 b12 = ld4(a.b12 + ncol), b13 = ld4(a.b13 + ncol);
    const f4 g4 = ld4(a.g3 + 4 * c32), be4 = ld4(a.be3 + 4 * c32);

#if TM_SETPRIO
    if (__builtin_amdgcn_readfirstlane(tm_tid()) >= 256) __builtin_amdgcn_s_setprio(1);    // (provably wave-uniform condition: s_setprio ignores EXEC)
#endif
    const TileRange tr = xcd_tile_range(a.T);
    int i = tr.begin;
    int cur = 0;
    f4 gai, gcj[3], e_cur[3], e_nxt[3];
    if (i < tr.end) {
        if (tid < TM_TILE) s_idx[0][tid] = a.E_idx[(size_t)i * TM_KS + tid];
        const float *src = a.hE + (size_t)i * TM_KS * TM_H;
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) e_cur[rb] = ld4(src + (eoff + 16 * rb * TM_H));
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) store_split<SP>(tE, 16 * rb + m, c4, e_cur[rb]);
        gai = ld4(a.P + (size_t)i * 256 + ucol);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) gcj[rb] = ld4(prow_of(s_idx[0][16 * rb + m], i));
        touch(gai);                                    // (so that the loop header needs no vmcnt wait of its own)
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) touch(gcj[rb]);
        __syncthreads();
    }
    // neighbour list of the NEXT tile: requested one whole iteration before it is published (after GEMM 1 of the iteration
    // that precedes its tile) — wavefront 0 used to sit on that load in front of the barrier the other seven were waiting at
    int nidx = -1;
    if (i < tr.end && tid < TM_TILE) nidx = a.E_idx[(size_t)(i + tr.step < tr.end ? i + tr.step : i) * TM_KS + tid];
    mark(-1);
#if TM_EDGE_UNROLL2
    // The loop body twice per trip with the two e-tile register sets swapping roles (e_a current / e_b next, then the reverse) and the
    // s_idx buffer index a constant: the loop-carried "e_cur = e_nxt" (12 v_mov per tile) disappears. Same operations, same order.
    f4 (&e_a)[3] = e_cur, (&e_b)[3] = e_nxt;
    auto tile_iter = [&](f4 (&e_cur)[3], f4 (&e_nxt)[3], const int cur) {
            float *tile_g = a.hE + (size_t)i * TM_KS * TM_H;
            const int inext = i + tr.step;
            const bool has_next = inext < tr.end;
            const int ipf = has_next ? inext : i;              // prefetch target (the last iteration re-reads its own tile)
            const int ipf2 = ipf + tr.step < tr.end ? ipf + tr.step : ipf;
            const int nidx_pub = nidx;                         // list of tile ipf, requested during the previous iteration
            {
                if (tid < TM_TILE) nidx = a.E_idx[(size_t)ipf2 * TM_KS + tid];
                const float *src = a.hE + (size_t)ipf * TM_KS * TM_H;         // wave-uniform base + per-thread offset
    #if TM_ABL_NOLOAD
    #pragma unroll
                for (int rb = 0; rb < 3; ++rb) e_nxt[rb] = e_cur[rb];
                (void)src;
    #else
    #pragma unroll
                for (int rb = 0; rb < 3; ++rb) e_nxt[rb] = ld4(src + (eoff + 16 * rb * TM_H));
    #endif
            }
            f4 acc[3][1];
    #pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = gai + gcj[rb];
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tE, w11, acc, lane);
            mark(0);
            {   // the three row blocks' GELUs as six independent chains, then the three splits
                f4 g[3];
    #pragma unroll
                for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int rb = 0; rb < 3; ++rb) store_split<SP>(tX, 16 * rb + m, c4, g[rb]);
            }
            if (tid < TM_TILE) s_idx[cur ^ 1][tid] = nidx_pub;
            mark(1);
            __syncthreads();
            mark(2);

            gai = ld4(a.P + (size_t)ipf * 256 + ucol);
    #pragma unroll
            for (int rb = 0; rb < 3; ++rb) gcj[rb] = ld4(prow_of(s_idx[cur ^ 1][16 * rb + m], ipf));
    #pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b12;
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tX, w12, acc, lane);
            mark(3);
            {
                f4 g[3];
    #pragma unroll
                for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int rb = 0; rb < 3; ++rb) store_split<SP>(tY, 16 * rb + m, c4, g[rb]);
            }
            mark(4);
            __syncthreads();
            mark(5);

    #pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b13;
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tY, w13, acc, lane);
            mark(6);
    #pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const f4 v = e_cur[rb] + acc[rb][0];                             // residual on the fp32 tile
                st4(tO + chunk_off(16 * rb + m, c4), v);
                #if TM_ABL_NOLN
                (void)q;
    #else
                row_stats_partial16(v, &s_stat[16 * rb + m][2 * wv], q);
    #endif
            }
            mark(7);
            __syncthreads();                                                     // tE free, tO + stats complete
            mark(8);

    #pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                store_split<SP>(tE, 16 * rb + m, c4, e_nxt[rb]);
                
            }
            touch(gai);                                    // the next tile's gathers have long arrived: take their vmcnt wait
    #pragma unroll
            for (int rb = 0; rb < 3; ++rb) touch(gcj[rb]); // here, in front of the stores below (see touch())
    #pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int row = 6 * wv + 2 * it + (lane >> 5);
                float mean = 0.f, rstd = 1.f;
    #if !TM_ABL_NOLN
                row_stats_finish8d(&s_stat[row][0], lane, mean, rstd);
    #endif
                // (x - mean) rstd g + be as y = x s + t with s = rstd g, t = be - mean s: three packed fmas / muls per half row
                const f4 x4 = ld4(tO + chunk_off(row, c32));
                const f2 s01 = f2{g4.x, g4.y} * rstd, s23 = f2{g4.z, g4.w} * rstd;
                const f2 t01 = __builtin_elementwise_fma(f2{-mean, -mean}, s01, f2{be4.x, be4.y});
                const f2 t23 = __builtin_elementwise_fma(f2{-mean, -mean}, s23, f2{be4.z, be4.w});
                const f2 y01 = __builtin_elementwise_fma(f2{x4.x, x4.y}, s01, t01), y23 = __builtin_elementwise_fma(f2{x4.z, x4.w}, s23, t23);
                const f4 y = f4{y01.x, y01.y, y23.x, y23.y};
                // rows without a neighbour keep the zeros the featurizer wrote: store zeros again (no divergent branch)
                st4(tile_g + (soff + 2 * it * TM_H), s_idx[cur][row] >= 0 ? y : f4{0.f, 0.f, 0.f, 0.f});
            }
            
            mark(9);
            __syncthreads();
            mark(10);
    };
    while (i < tr.end) {
        tile_iter(e_a, e_b, 0);
        i += tr.step;
        if (!(i < tr.end)) break;
        tile_iter(e_b, e_a, 1);
        i += tr.step;
    }
#else
    for (; i < tr.end; i += tr.step) {
        float *tile_g = a.hE + (size_t)i * TM_KS * TM_H;
        const int inext = i + tr.step;
        const bool has_next = inext < tr.end;
        const int ipf = has_next ? inext : i;              // prefetch target (the last iteration re-reads its own tile)
        const int ipf2 = ipf + tr.step < tr.end ? ipf + tr.step : ipf;
        const int nidx_pub = nidx;                         // list of tile ipf, requested during the previous iteration
        {
            if (tid < TM_TILE) nidx = a.E_idx[(size_t)ipf2 * TM_KS + tid];
            const float *src = a.hE + (size_t)ipf * TM_KS * TM_H;         // wave-uniform base + per-thread offset
#if TM_ABL_NOLOAD
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) e_nxt[rb] = e_cur[rb];
            (void)src;
#else
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) e_nxt[rb] = ld4(src + (eoff + 16 * rb * TM_H));
#endif
        }
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = gai + gcj[rb];
        mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tE, w11, acc, lane);
        mark(0);
        {   // the three row blocks' GELUs as six independent chains, then the three splits
            f4 g[3];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) store_split<SP>(tX, 16 * rb + m, c4, g[rb]);
        }
        if (tid < TM_TILE) s_idx[cur ^ 1][tid] = nidx_pub;
        mark(1);
        __syncthreads();
        mark(2);

        gai = ld4(a.P + (size_t)ipf * 256 + ucol);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) gcj[rb] = ld4(prow_of(s_idx[cur ^ 1][16 * rb + m], ipf));
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b12;
        mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tX, w12, acc, lane);
        mark(3);
        {
            f4 g[3];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) store_split<SP>(tY, 16 * rb + m, c4, g[rb]);
        }
        mark(4);
        __syncthreads();
        mark(5);

#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b13;
        mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tY, w13, acc, lane);
        mark(6);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const f4 v = e_cur[rb] + acc[rb][0];                             // residual on the fp32 tile
            st4(tO + chunk_off(16 * rb + m, c4), v);
            #if TM_ABL_NOLN
            (void)q;
#else
            row_stats_partial16(v, &s_stat[16 * rb + m][2 * wv], q);
#endif
        }
        mark(7);
        __syncthreads();                                                     // tE free, tO + stats complete
        mark(8);

#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            store_split<SP>(tE, 16 * rb + m, c4, e_nxt[rb]);
            e_cur[rb] = e_nxt[rb];
        }
        touch(gai);                                    // the next tile's gathers have long arrived: take their vmcnt wait
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) touch(gcj[rb]); // here, in front of the stores below (see touch())
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int row = 6 * wv + 2 * it + (lane >> 5);
            float mean = 0.f, rstd = 1.f;
#if !TM_ABL_NOLN
            row_stats_finish8d(&s_stat[row][0], lane, mean, rstd);
#endif
            // (x - mean) rstd g + be as y = x s + t with s = rstd g, t = be - mean s: three packed fmas / muls per half row
            const f4 x4 = ld4(tO + chunk_off(row, c32));
            const f2 s01 = f2{g4.x, g4.y} * rstd, s23 = f2{g4.z, g4.w} * rstd;
            const f2 t01 = __builtin_elementwise_fma(f2{-mean, -mean}, s01, f2{be4.x, be4.y});
            const f2 t23 = __builtin_elementwise_fma(f2{-mean, -mean}, s23, f2{be4.z, be4.w});
            const f2 y01 = __builtin_elementwise_fma(f2{x4.x, x4.y}, s01, t01), y23 = __builtin_elementwise_fma(f2{x4.z, x4.w}, s23, t23);
            const f4 y = f4{y01.x, y01.y, y23.x, y23.y};
            // rows without a neighbour keep the zeros the featurizer wrote: store zeros again (no divergent branch)
            st4(tile_g + (soff + 2 * it * TM_H), s_idx[cur][row] >= 0 ? y : f4{0.f, 0.f, 0.f, 0.f});
        }
        cur ^= 1;
        mark(9);
        __syncthreads();
        mark(10);
    }
#endif
}

int launch_enc_edge_split(int mode, const EncW &e, const float *P, float *hE, const int32_t *E_idx, int64_t T, hipStream_t st) {
    const bool h2 = mode == TM_MM_F16X2;
    EdgeArgsB a{e.W11 + 128, e.W12, e.b12, e.W13, e.b13, e.norm3_w, e.norm3_b, P, hE, E_idx, (int)T,
                h2 ? tm_find_wimg(e.W11 + 128) : nullptr, h2 ? tm_find_wimg(e.W12) : nullptr, h2 ? tm_find_wimg(e.W13) : nullptr};
    const int64_t cap = tm_num_cus();
    const int grid = (int)(T < cap ? T : cap);
    if (mode == TM_MM_BF16X3) enc_edge8_split_kernel<SplitBF3><<<grid, 512, 0, st>>>(a);
    else {
#ifdef TMPNN_DEBUG_BUILD
        static const bool prof = TM_DBG_FLAG("TMPNN_EDGE_PROF", false);
#else
        constexpr bool prof = false;
#endif
        if (prof) {                                  // debug build: phase timing of workgroup 0 (synchronises!)
#ifdef TMPNN_DEBUG_BUILD
            static unsigned long long *d_prof = nullptr;
            if (!d_prof) (void)hipMalloc(&d_prof, 16 * sizeof(unsigned long long));
            (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), st);
            enc_edge8_rp_kernel<SplitH2, true, false><<<grid, 512, 0, st>>>(a, d_prof);
            unsigned long long h[16];
            (void)hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "enc_edge phases (cycles, wg 0): gemm1 %llu gelu+split %llu bar %llu gather+gemm2 %llu gelu+split %llu bar %llu gemm3 %llu resid+stats %llu bar %llu split+ln+store %llu bar %llu\n",
                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10]);
#endif
        } else if (T < ((int64_t)1 << 22)) {
            enc_edge8_rp_kernel<SplitH2, false, true><<<grid, 512, 0, st>>>(a);      // projection table < 4 GB: 32-bit gather offsets
        } else {
            enc_edge8_rp_kernel<SplitH2><<<grid, 512, 0, st>>>(a);
        }
    }
    return tm_check_launch("enc_edge_split");
}


// ------------------------------------------------------------------------------------------------
// message kernels, split-precision form (8 wavefronts, 1 workgroup per CU, next tile prefetched through registers).
// Same arithmetic as msg_kernel (tmpnn_layers.hip): Ssum_i = sum_k ma_ik gelu(W2 gelu(pre_ik) + b2).
// ------------------------------------------------------------------------------------------------
struct MsgArgsB {
    const float *W1e; int ld1;
    const float *W2, *b2, *P;
    const float *hE;
    const int32_t *E_idx;
    const float *mask;
    float *Ssum, *cnt;
    int T;
    const char *img1, *img2;                // fragment images of W1e / W2 (f16x2 only) or null
};

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
#if TM_MSG_PFD == 2
    f4 e_far[3];                                       // the tile after e_nxt's
#endif
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

#if TM_SETPRIO
    if (__builtin_amdgcn_readfirstlane(tm_tid()) >= 256) __builtin_amdgcn_s_setprio(1);
#endif
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
#if TM_MSG_PFD == 2
        fetch_into(e_far, i + 2 * tr.step < tr.end ? i + 2 * tr.step : i);
#endif
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
#if TM_ABL_NOLOAD
#elif TM_MSG_PFD == 2
#pragma unroll
        for (int it = 0; it < 3; ++it) e_nxt[it] = e_far[it];   // (register renaming: the planes just took e_nxt)
        fetch_into(e_far, i + 3 * tr.step < tr.end ? i + 3 * tr.step : ipf);
#else
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
        if (TM_MSG_TOUCH) {                                      // take the gathers' vmcnt wait before any store is issued (see touch())
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

// ------------------------------------------------------------------------------------------------
// Small launches (T <= #CUs: one tile per workgroup — a single protein, a handful of short ones): the edge update of
// encoder layer l and the message pass of the NEXT layer (encoder l+1, or decoder 0 after the last encoder layer) as ONE
// launch. Both need only this residue's edge tile plus node projections that node_update(l) has already written, so there is
// no grid-wide dependency between them; a launch costs 2.5 us of dispatch + a prologue even when it does nothing
// (tools/gap_probe.py), and the fresh LayerNorm'd tile is in registers in exactly the row layout the message pass splits from.
// The five weight fragments do not have to be resident together here (nothing persists across tiles): the message weights are
// loaded into the registers the edge weights leave. Arithmetic = enc_edge8_rp_kernel followed by msg8_rp_kernel, operation for
// operation (the same GEMM step order, the same epilogue expressions): results are BIT-IDENTICAL to the two-launch path, so a
// protein's numbers do not depend on the batch it is in (tests: test_small_launch_fused_forms_are_bit_identical).
// ------------------------------------------------------------------------------------------------
template <typename SP, bool DEC>
__global__ __launch_bounds__(512, 2) void edge_msg_fused_kernel(EdgeArgsB a, MsgArgsB b) {
    constexpr int TILEB = SP::NP * SPLIT_PLANE_BYTES;
    static_assert(TILEB >= TM_TILE * TM_H * 4, "the fp32 LayerNorm tile is aliased on the x planes");
    __shared__ __attribute__((aligned(16))) char tE[TILEB];
    __shared__ __attribute__((aligned(16))) char tX[TILEB];              // x planes; the fp32 LayerNorm input; the message pass's tA
    // GEMM 2's output planes live where the e planes were: GEMM 1 was their last reader (every wavefront is past the barrier behind
    // it), the next tile's e planes are written only behind the barrier that follows GEMM 3. Two plane tiles instead of three:
    // 53 KB of LDS, every LDS offset below 64 KB (an offset above costs an address VGPR + a v_or each: 10 VALU per tile).
#if TM_EDGE_Y_ALIAS
    char *const tY = tE;
#else
    __shared__ __attribute__((aligned(16))) char tY[TILEB];
#endif
    __shared__ __attribute__((aligned(16))) float s_stat[TM_TILE][TM_STAT_LD];
    __shared__ int s_idx[TM_TILE];
    __shared__ float s_ma[TM_TILE];
    float *tO = reinterpret_cast<float *>(tX);
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const int c32 = lane & 31;
    const unsigned ucol = (unsigned)ncol;
    const unsigned eoff = (unsigned)(m * TM_H + ncol);
    const unsigned soff = (unsigned)((6 * wv + (lane >> 5)) * TM_H + 4 * c32);

    for (int i = tm_bid(); i < a.T; i += tm_nblk()) {
        // ---- edge update of this tile (enc_edge8_rp_kernel) ------------------------------------------
        f4 e_cur[3], yrow[3];
        f4 g0, gj[3];                                    // the message pass's node terms: requested with the edge update's (same list)
        float mi, nma = 0.f;
        // The five weight fragments are a software pipeline through TWO register sets (64 VGPRs), each requested one GEMM phase
        // ahead of its use, into the set the previous GEMM has just finished with: fa = W11 -> W13 -> W2, fb = W12 -> W1.
        // (All five resident, or the message pair requested early, spills — and a scratch reload's vmcnt wait drains every
        //  prefetch in flight: 18.4 us per launch against 15.8.)
        WFragS<SP> fa[1][4], fb[1][4];
        f4 bias2;
        {
            load_wfrag_auto<SP>(a.img11, a.W11e, 384, wv, lane, fa[0]);
            load_wfrag_auto<SP>(a.img12, a.W12, TM_H, wv, lane, fb[0]);
            const f4 b12 = ld4(a.b12 + ncol), b13 = ld4(a.b13 + ncol);
            const f4 g4 = ld4(a.g3 + 4 * c32), be4 = ld4(a.be3 + 4 * c32);
            if (tid < TM_TILE) s_idx[tid] = a.E_idx[(size_t)i * TM_KS + tid];
            float *tile_g = a.hE + (size_t)i * TM_KS * TM_H;
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) e_cur[rb] = ld4(tile_g + (eoff + 16 * rb * TM_H));
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) store_split<SP>(tE, 16 * rb + m, c4, e_cur[rb]);
            f4 gai = ld4(a.P + (size_t)i * 256 + ucol), gcj[3];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int j = s_idx[16 * rb + m];
                gcj[rb] = ld4(a.P + (size_t)(j < 0 ? i : j) * 256 + 128 + ncol);
            }
            g0 = ld4(b.P + (size_t)i * 256 + ucol);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int j0 = s_idx[16 * rb + m];
                gj[rb] = ld4(b.P + (size_t)(j0 < 0 ? i : j0) * 256 + 128 + ncol);
            }
            mi = b.mask[i];
            if (tid < TM_TILE) {                          // (only REQUESTED here; the product is formed in the message phase — a
                const int j = s_idx[tid];                 //  use here would wait for every load above, in front of GEMM 1)
                nma = b.mask[j < 0 ? i : j];
            }
            __syncthreads();
            f4 acc[3][1];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = gai + gcj[rb];
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tE, fa, acc, lane);
            __builtin_amdgcn_sched_barrier(0);
            load_wfrag_auto<SP>(a.img13, a.W13, TM_H, wv, lane, fa[0]);          // W11 is done with: W13 for GEMM 3
            {
                f4 g[3];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) store_split<SP>(tX, 16 * rb + m, c4, g[rb]);
            }
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b12;
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tX, fb, acc, lane);
            __builtin_amdgcn_sched_barrier(0);
            load_wfrag_auto<SP>(b.img1, b.W1e, b.ld1, wv, lane, fb[0]);          // W12 is done with: the message pass's W1
            {
                f4 g[3];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) store_split<SP>(tY, 16 * rb + m, c4, g[rb]);
            }
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b13;
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tY, fa, acc, lane);
            __builtin_amdgcn_sched_barrier(0);
            load_wfrag_auto<SP>(b.img2, b.W2, TM_H, wv, lane, fa[0]);            // W13 is done with: the message pass's W2
            bias2 = ld4(b.b2 + ncol);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const f4 v = e_cur[rb] + acc[rb][0];                             // residual on the fp32 tile
                st4(tO + chunk_off(16 * rb + m, c4), v);
                row_stats_partial16(v, &s_stat[16 * rb + m][2 * wv], q);
            }
            __syncthreads();                                                     // tE free, tO + stats complete
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int row = 6 * wv + 2 * it + (lane >> 5);
                float mean = 0.f, rstd = 1.f;
                row_stats_finish8d(&s_stat[row][0], lane, mean, rstd);
                const f4 x4 = ld4(tO + chunk_off(row, c32));
                const f2 s01 = f2{g4.x, g4.y} * rstd, s23 = f2{g4.z, g4.w} * rstd;
                const f2 t01 = __builtin_elementwise_fma(f2{-mean, -mean}, s01, f2{be4.x, be4.y});
                const f2 t23 = __builtin_elementwise_fma(f2{-mean, -mean}, s23, f2{be4.z, be4.w});
                const f2 y01 = __builtin_elementwise_fma(f2{x4.x, x4.y}, s01, t01), y23 = __builtin_elementwise_fma(f2{x4.z, x4.w}, s23, t23);
                const f4 y = f4{y01.x, y01.y, y23.x, y23.y};
                yrow[it] = s_idx[row] >= 0 ? y : f4{0.f, 0.f, 0.f, 0.f};         // rows without a neighbour stay zero
                st4(tile_g + (soff + 2 * it * TM_H), yrow[it]);                  // the later kernels read the updated tile from HBM
            }
        }
        // ---- message pass of the next layer on the SAME tile (msg8_rp_kernel) ------------------------------
        {
            if (tid < TM_TILE) s_ma[tid] = s_idx[tid] < 0 ? 0.f : (DEC ? 1.f : mi * nma);
            const int prow = 6 * wv + (lane >> 5), pc = lane & 31;              // the row layout the tile was just produced in
#pragma unroll
            for (int it = 0; it < 3; ++it) store_split<SP>(tE, prow + 2 * it, pc, yrow[it]);
            __syncthreads();                                                     // e planes + s_ma complete; tO (= tA) consumed
            f4 acc[3][1];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = DEC ? gj[rb] : g0 + gj[rb];
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_MSG_PF>(tE, fb, acc, lane);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                f4 v = acc[rb][0];
                if (DEC) v = g0 + mi * v;
                store_split<SP>(tX, 16 * rb + m, c4, gelu4(v));
            }
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = bias2;
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_MSG_PF>(tX, fa, acc, lane);
            f4 tot = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const float ma = s_ma[16 * rb + m];
                const f4 g = gelu4(acc[rb][0]);
                tot = f4{__builtin_fmaf(g.x, ma, tot.x), __builtin_fmaf(g.y, ma, tot.y), __builtin_fmaf(g.z, ma, tot.z), __builtin_fmaf(g.w, ma, tot.w)};
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = tot[c];
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
                tot[c] = x;
            }
            if (m == 15) st4(b.Ssum + (size_t)i * TM_H + ucol, tot);
            if (wv == 2) {
                float c = lane < TM_TILE ? s_ma[lane] : 0.f;
#define TM_DPP_ADD(ctrl, row_mask, bc)                                                                  \
                c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), ctrl, row_mask, 0xf, bc));
                TM_DPP_ADD(0x111, 0xf, true)
                TM_DPP_ADD(0x112, 0xf, true)
                TM_DPP_ADD(0x114, 0xf, true)
                TM_DPP_ADD(0x118, 0xf, true)
                TM_DPP_ADD(0x142, 0xa, false)
                TM_DPP_ADD(0x143, 0xc, false)
#undef TM_DPP_ADD
                if (lane == 63) b.cnt[i] = c;
            }
            __syncthreads();                                                     // (a further tile of this workgroup reuses every buffer)
        }
    }
}

// The fused form is used when every workgroup has at most one tile and the fragment images exist (f16x2 handles).
bool edge_msg_fusable(int mode, int64_t T) { return mode == TM_MM_F16X2 && T > 0 && T <= (int64_t)tm_num_cus(); }

int launch_edge_msg_fused(const EncW &e, const float *P_edge, float *hE, const int32_t *E_idx, bool dec, const float *W1e, int ld1,
                          const float *W2, const float *b2, const float *P_msg, const float *mask, int64_t T, float *Ssum, float *cnt,
                          hipStream_t st) {
    EdgeArgsB a{e.W11 + 128, e.W12, e.b12, e.W13, e.b13, e.norm3_w, e.norm3_b, P_edge, hE, E_idx, (int)T,
                tm_find_wimg(e.W11 + 128), tm_find_wimg(e.W12), tm_find_wimg(e.W13)};
    MsgArgsB b{W1e, ld1, W2, b2, P_msg, hE, E_idx, mask, Ssum, cnt, (int)T, tm_find_wimg(W1e), tm_find_wimg(W2)};
    const int64_t cap = tm_num_cus();
    const int grid = (int)(T < cap ? T : cap);
    tm_prof_begin("edge_msg_fused", st);
    if (dec) edge_msg_fused_kernel<SplitH2, true><<<grid, 512, 0, st>>>(a, b);
    else edge_msg_fused_kernel<SplitH2, false><<<grid, 512, 0, st>>>(a, b);
    tm_prof_end(st);
    return tm_check_launch("edge_msg_fused");
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

// ------------------------------------------------------------------------------------------------
// node_update, 8-wavefront f16x2 form (default): one workgroup per CU, up to 64 residues per tile, 16 output columns per
// wavefront. A tile runs 9..13 dependent GEMMs whose weights stream from L2: the raw fp32 fragment of GEMM u+1 (32 VGPRs)
// is requested before the MFMAs of GEMM u and split into f16 planes after them, so no GEMM waits on an L2 round trip
// (the 4-wavefront form above does, 13 times per tile); the taller tile halves the weight traffic per residue.
// ------------------------------------------------------------------------------------------------
template <typename SP, int NRB, bool IMG, bool PROF = false>
__global__ __launch_bounds__(512, 2) void node_update8_split_kernel(NodeArgs a, unsigned long long *prof = nullptr) {
    int n_mark = 0;
    auto mark = [&]() {                 // TMPNN_NODE_PROF=1: cycle stamps of thread 0 of workgroup 0 at every stage boundary (first tile)
        if (PROF && tm_bid() == 0 && tm_tid() == 0 && n_mark < 32) prof[n_mark++] = __builtin_readcyclecounter();
    };
    mark();
    constexpr int ROWS = 16 * NRB, PLT = SP::NP * ROWS * 256;
    static_assert(PLT >= ROWS * TM_H * 4, "the fp32 LayerNorm-2 input is aliased on the plane tile pA");
    __shared__ __attribute__((aligned(16))) char pA[PLT];
    __shared__ __attribute__((aligned(16))) char pB[PLT];
    __shared__ __attribute__((aligned(16))) float tB[ROWS * TM_H];
    // Every small operand of the tile comes from LDS: the layer's bias / LayerNorm vectors and the sequence tables once per
    // workgroup, the tile's own rows (old state into tB, neighbour counts, masks, table indices) with the tile's first loads.
    // gfx9 retires loads and stores in order: a bias fetched from global memory at an accumulator initialisation waited for
    // the 64 KB weight fragment requested just before it and for the previous unit's 32 KB of stores — stage timers showed
    // 5 k cycles per W_in unit against 2.3 k per W_out unit (no bias) and 9 k per projection half (stores + bias).
    enum { P_B3 = 0, P_BOUT = 128, P_BIN = 256, P_N1W = 768, P_N1B = 896, P_N2W = 1024, P_N2B = 1152, P_BA = 1280, P_END = 1536 };
    __shared__ __attribute__((aligned(16))) float s_par[P_END];
    __shared__ __attribute__((aligned(16))) float s_add[2][TMPNN_VOCAB * TM_H];
    __shared__ float s_cnt[ROWS], s_mask[ROWS];
    __shared__ int s_aidx[2][ROWS];
    float *tA = reinterpret_cast<float *>(pA);
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int c32 = lane & 31, hw = tid >> 5;                  // half-wavefront index: rows hw*NRB .. hw*NRB + NRB - 1
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const int n_tiles = (a.T + ROWS - 1) / ROWS;
    const bool has0 = a.proj[0].P != nullptr, has1 = a.proj[1].P != nullptr;

    // GEMM units of a tile: 0 = W3; 1 + 2c = W_in chunk c, 2 + 2c = W_out chunk c; 9 / 10 = projection 0 (A / C half);
    // 11 / 12 = projection 1. src(u) = this lane's fragment row: W[(n0 + m) * ld + k0 + 8 q ...]
    auto src = [&](int u) -> const float * {
        const size_t r = (size_t)(16 * wv + m);
        if (u == 0) return a.W3 + r * TM_H + 8 * q;
        if (u <= 8) {
            const int c = (u - 1) >> 1;
            return ((u - 1) & 1) ? a.Wout + r * 512 + 128 * c + 8 * q : a.Win + (r + 128 * c) * TM_H + 8 * q;
        }
        const ProjSpec &ps = a.proj[(u - 9) >> 1];
        return ((u - 9) & 1) ? ps.Wc + r * ps.ldc + 8 * q : ps.Wa + r * ps.lda + 8 * q;
    };
    // With pre-built fragment images (NodeArgs::img, built by tmpnn_weights_create) a unit's fragment is 8 coalesced 1 KB
    // loads of ready-made f16 planes; without them (standalone callers) it is gathered from 16 fp32 rows per load and split
    // on the fly. Measured (MI355X): 22.9 vs 30.3 us per launch on a single L=256 protein — the strided gathers ran at a
    // third of the L2 -> CU fill rate and every one of the 9-13 dependent GEMM units of a tile waited for them.
    f4 raw[8];
    auto issue = [&](int u) {
        if constexpr (IMG) {
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
    const int first_proj = has0 ? 9 : 11;                       // first projection unit, if any

    int tile = tm_bid();
    if (tile >= n_tiles) return;
    {   // every load unconditional and requested before the first LDS write (a load under a branch is waited for at the join:
        // written the obvious way this block was seven dependent round trips, 9 k cycles)
        const int t7 = tid & 127;
        const bool hp0 = a.proj[0].P != nullptr, hp1 = a.proj[1].P != nullptr;
        const float *ba0 = hp0 ? a.proj[0].ba : a.b3, *ba1 = hp1 ? a.proj[1].ba : a.b3;
        const bool ha0 = hp0 && a.proj[0].add_tab != nullptr, ha1 = hp1 && a.proj[1].add_tab != nullptr;
        const float *at0 = ha0 ? a.proj[0].add_tab : a.bin, *at1 = ha1 ? a.proj[1].add_tab : a.bin;     // (dummies: any 512 valid floats)
        const float vbin = a.bin[tid];
        const float v6[8] = {a.b3[t7], a.bout[t7], a.n1w[t7], a.n1b[t7], a.n2w[t7], a.n2b[t7], ba0[t7], ba1[t7]};
        constexpr int NADD = (TMPNN_VOCAB * TM_H + 511) / 512;
        float va[2][NADD];
#pragma unroll
        for (int j = 0; j < NADD; ++j) {
            const int e = tid + 512 * j;
            va[0][j] = at0[ha0 && e < TMPNN_VOCAB * TM_H ? e : tid];
            va[1][j] = at1[ha1 && e < TMPNN_VOCAB * TM_H ? e : tid];
        }
        s_par[P_BIN + tid] = vbin;
        if (tid < 128) {
            s_par[P_B3 + tid] = v6[0];
            s_par[P_BOUT + tid] = v6[1];
            s_par[P_N1W + tid] = v6[2];
            s_par[P_N1B + tid] = v6[3];
            s_par[P_N2W + tid] = v6[4];
            s_par[P_N2B + tid] = v6[5];
            s_par[P_BA + tid] = v6[6];
            s_par[P_BA + 128 + tid] = v6[7];
        }
#pragma unroll
        for (int j = 0; j < NADD; ++j) {
            const int e = tid + 512 * j;
            if (e < TMPNN_VOCAB * TM_H) {
                s_add[0][e] = va[0][j];
                s_add[1][e] = va[1][j];
            }
        }
    }
    mark();
    issue(0);
    for (; tile < n_tiles; tile += tm_nblk()) {
        const int r0 = tile * ROWS;
        {   // aggregated messages -> planes, old state -> tB: all 2 NRB row chunks of this thread requested before the first is used
            static_assert(ROWS * 32 == 512 * NRB, "one 16-byte chunk of NRB rows per thread");
            f4 v[NRB], hvv[NRB];
#pragma unroll
            for (int it = 0; it < NRB; ++it) {
                const int row = 16 * it + (tid >> 5), c = tid & 31;
                const size_t g = (size_t)(r0 + row < a.T ? r0 + row : r0) * TM_H + 4 * c;   // (rows past T: a valid row, zeroed below)
                v[it] = ld4(a.Ssum + g);
                hvv[it] = ld4(a.h_in + g);
            }
#pragma unroll
            for (int it = 0; it < NRB; ++it) {
                const int row = 16 * it + (tid >> 5), c = tid & 31;
                const bool ok = r0 + row < a.T;
                const f4 z4 = f4{0.f, 0.f, 0.f, 0.f};
                store_split<SP, ROWS>(pA, row, c, ok ? v[it] : z4);
                st4(tB + chunk_off(row, c), ok ? hvv[it] : z4);
            }
        }
        mark();
        if (tid < ROWS) {
            const bool ok = r0 + tid < a.T;
            const int g = ok ? r0 + tid : r0;
            const float cv = a.cnt[g], mv = a.mask[g];
            s_cnt[tid] = ok ? cv : 0.f;
            s_mask[tid] = ok ? mv : 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (a.proj[k].P != nullptr && a.proj[k].add_tab != nullptr) s_aidx[k][tid] = ok ? a.proj[k].add_idx[g] : 0;
        }
        __syncthreads();
        mark();

        f4 acc[NRB][1];
        split_raw();
        issue(1);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = f4{0.f, 0.f, 0.f, 0.f};
        mma_tile_split<SP, 4, 1, NRB, ROWS, 256, 4, 0, true, TM_NODE_PF>(pA, wf, acc, lane);
        {
            const f4 b3 = ld4(s_par + P_B3 + ncol);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                float *p = tB + chunk_off(16 * rb + m, c4);      // holds the old state of (row, these 4 columns): this thread's own slot
                const float c = s_cnt[16 * rb + m];
                const f4 hv = ld4(p);
                const f4 dh = fma4s(c, b3, acc[rb][0]) / 30.0f;
                st4(p, hv + dh);
            }
        }
        __syncthreads();
        mark();
        {   // LN1: fp32 in place (the FFN residual) + planes (the FFN input)
            const f4 g4 = ld4(s_par + P_N1W + 4 * c32), b4 = ld4(s_par + P_N1B + 4 * c32);
#pragma unroll
            for (int it = 0; it < NRB; ++it) {
                const int row = NRB * hw + it;
                float *p = tB + chunk_off(row, c32);
                const f4 y = layer_norm_row(ld4(p), g4, b4);
                st4(p, y);
                store_split<SP, ROWS>(pB, row, c32, y);
            }
        }
        __syncthreads();
        mark();

        f4 out[NRB][1];
        {
            const f4 b = ld4(s_par + P_BOUT + ncol);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) out[rb][0] = b;
        }
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {           // FFN hidden 512 in four 128-wide chunks
            split_raw();                        // W_in chunk c
            issue(2 + 2 * c);
            {
                const f4 b = ld4(s_par + P_BIN + 128 * c + ncol);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b;
            }
            mma_tile_split<SP, 4, 1, NRB, ROWS, 256, 4, 0, true, TM_NODE_PF>(pB, wf, acc, lane);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) store_split<SP, ROWS>(pA, 16 * rb + m, c4, gelu4(acc[rb][0]));
            __syncthreads();
            mark();
            split_raw();                        // W_out chunk c
            if (c < 3) issue(3 + 2 * c);
            else if (has0 || has1) issue(first_proj);
            else if (tile + (int)tm_nblk() < n_tiles) issue(0);
            mma_tile_split<SP, 4, 1, NRB, ROWS, 256, 4, 0, true, TM_NODE_PF>(pA, wf, out, lane);
            __syncthreads();
            mark();
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int off = chunk_off(16 * rb + m, c4);
            st4(tA + off, ld4(tB + off) + out[rb][0]);                       // tA aliases pA: every wavefront is past its last read
        }
        __syncthreads();
        {   // LN2, mask, coalesced store; the new state goes into the planes pB for the projections
            const f4 g4 = ld4(s_par + P_N2W + 4 * c32), b4 = ld4(s_par + P_N2B + 4 * c32);
#pragma unroll
            for (int it = 0; it < NRB; ++it) {
                const int row = NRB * hw + it;
                const int grow = r0 + row;
                f4 y = layer_norm_row(ld4(tA + chunk_off(row, c32)), g4, b4);
                y = grow < a.T ? y * s_mask[row] : f4{0.f, 0.f, 0.f, 0.f};
                store_split<SP, ROWS>(pB, row, c32, y);
                if (grow < a.T) st4(a.h_out + (size_t)grow * TM_H + 4 * c32, y);
            }
        }
        __syncthreads();
        mark();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const ProjSpec &ps = a.proj[k];
            if (ps.P == nullptr) continue;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                split_raw();
                // next unit: the C half, the other projection, or W3 of this workgroup's next tile
                if (!half) issue(10 + 2 * k);
                else if (k == 0 && has1) issue(11);
                else if (tile + (int)tm_nblk() < n_tiles) issue(0);
                {
                    const f4 b = half ? f4{0.f, 0.f, 0.f, 0.f} : ld4(s_par + P_BA + 128 * k + ncol);
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b;
                }
                mma_tile_split<SP, 4, 1, NRB, ROWS, 256, 4, 0, true, TM_NODE_PF>(pB, wf, acc, lane);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const int row = r0 + 16 * rb + m;
                    if (row < a.T) {
                        const float *add = half && ps.add_tab ? s_add[k] + s_aidx[k][16 * rb + m] * TM_H : nullptr;
                        st4(ps.P + (size_t)row * 256 + 128 * half + ncol, add ? ld4(add + ncol) + acc[rb][0] : acc[rb][0]);
                    }
                }
                mark();
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// node_update for FEW residues (every workgroup has exactly one 16-row tile: T <= 16 x #CUs — a single protein or a small
// batch, the latency case). A tile is a chain of 9..13 dependent GEMM units whose weights come from L2; with one unit
// requested ahead (the form above) every unit waits out most of an L2 round trip (~0.6 us x 13). Here the fragment images of
// the next D units are in flight at any time, in a ring of D + 1 register slots that the MFMAs read in place (a 16-row tile
// needs few other VGPRs), and every small operand (biases, LayerNorm parameters, the tile's own rows) is requested BEFORE the
// ring is primed: gfx9's vmcnt retires in order, a later wait for a small load would drain the whole ring.
// Arithmetic and its order are those of node_update8_split_kernel (bit-identical results).
// ------------------------------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int NPROJ, int D, bool PROF = false>
__global__ __launch_bounds__(512) void node_update8_deep_kernel(NodeArgs a, unsigned long long *prof = nullptr) {
    using SP = SplitH2;
    int n_mark = 0;
    auto mark = [&]() {                 // TMPNN_NODE_PROF=1: cycle stamps of thread 0 of workgroup 0 at every stage boundary
        if (PROF && tm_bid() == 0 && tm_tid() == TM_PROF_TID) prof[n_mark++] = __builtin_readcyclecounter();
    };
    mark();
    kernarg_warm<sizeof(NodeArgs)>();
    constexpr int ROWS = 16, PLT = SP::NP * ROWS * 256, NPOS = 9 + 2 * NPROJ, NS = D + 1;
    static_assert(PLT >= ROWS * TM_H * 4, "the fp32 LayerNorm-2 input is aliased on the plane tile pA");
    __shared__ __attribute__((aligned(16))) char pA[PLT];
    __shared__ __attribute__((aligned(16))) char pB[PLT];
    __shared__ __attribute__((aligned(16))) float tB[ROWS * TM_H];
    float *tA = reinterpret_cast<float *>(pA);
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int c32 = lane & 31, hw = tid >> 5;                  // half-wavefront hw owns row hw in the row phases
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const int r0 = tm_bid() * ROWS;                           // the launcher starts exactly ceil(T / 16) workgroups
    // The launcher compacts the projections (the NPROJ present ones first, their images in img[9..]): every kernel argument
    // is then read at a fixed offset and the scalar loads form one cluster (a dependent second round trip to the freshly
    // written argument buffer costs ~0.5 us).
    constexpr int pk[2] = {0, 1};
    // unit of position p: 0 = W3; 1 + 2c / 2 + 2c = W_in / W_out chunk c; then the A and C halves of the projections
    auto unit_at = [&](int p) { return p; };

    // ---- small operands first. Every load is unconditional (rows past T are clamped to the tile's first row and masked
    // afterwards, absent tables are replaced by a valid dummy) so that hipcc keeps the scalar argument loads in one cluster and
    // the vector loads back to back: conditional loads became a chain of s_load / s_waitcnt / branch blocks (2 us of the tile).
    const f4 z4 = f4{0.f, 0.f, 0.f, 0.f};
    const int row_m = r0 + m, grow = r0 + hw;
    const bool ok_m = row_m < a.T, ok_h = grow < a.T;
    const int row_c = ok_m ? row_m : r0, grow_c = ok_h ? grow : r0;
    const f4 sv_raw = ld4(a.Ssum + (size_t)grow_c * TM_H + 4 * c32);             // ROWS * 32 chunks = one per thread
    const f4 hv_raw = ld4(a.h_in + (size_t)row_c * TM_H + ncol);
    const float cnt_raw = a.cnt[row_c], mk_raw = a.mask[grow_c];
    bool has_add[2] = {false, false};
    int add_row[2] = {0, 0};
    f4 pb[2] = {z4, z4}, padd[2] = {z4, z4};
#pragma unroll
    for (int k = 0; k < NPROJ; ++k) {
        const ProjSpec &ps = a.proj[pk[k]];
        has_add[k] = ps.add_tab != nullptr;
        add_row[k] = (has_add[k] ? ps.add_idx : reinterpret_cast<const int32_t *>(a.cnt))[row_c];
        pb[k] = ld4(ps.ba + ncol);
    }
    const f4 b3 = ld4(a.b3 + ncol), bout = ld4(a.bout + ncol);
    f4 bin[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bin[c] = ld4(a.bin + 128 * c + ncol);
    const f4 g1 = ld4(a.n1w + 4 * c32), be1 = ld4(a.n1b + 4 * c32), g2 = ld4(a.n2w + 4 * c32), be2 = ld4(a.n2b + 4 * c32);

    // ---- the ring
    WFragS<SP> ring[NS][1][4];
    auto issue = [&](auto P) {
        constexpr int p = decltype(P)::value;
        if constexpr (p < NPOS) {
            const char *src = a.img[unit_at(p)] + (size_t)wv * 8192 + lane * 16;
            __builtin_amdgcn_sched_barrier(0);                  // the loads stay HERE: hoisted, they would need more slots
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ring[p % NS][0][c].p[0] = *reinterpret_cast<const u4 *>(src + 2048 * c);
                ring[p % NS][0][c].p[1] = *reinterpret_cast<const u4 *>(src + 2048 * c + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_for<0, D>(issue);
    // (the masks use the loaded values: in front of the ring they would make it wait for them)
    const f4 sv = ok_h ? sv_raw : z4, hv = ok_m ? hv_raw : z4;
    const float cnt = ok_m ? cnt_raw : 0.f, mk = ok_h ? mk_raw : 0.f;
    // the one dependent gather (index -> table row) goes AFTER the ring: its index is older than the ring's loads, so waiting
    // for it drains nothing, and its rows are not needed before the last GEMM unit
#pragma unroll
    for (int k = 0; k < NPROJ; ++k) {
        asm volatile("" : "+v"(add_row[k]));                    // first use of the index HERE (its vmcnt wait comes with it)
        const f4 t = ld4((has_add[k] ? a.proj[pk[k]].add_tab + (size_t)add_row[k] * TM_H : a.b3) + ncol);
        padd[k] = has_add[k] && ok_m ? t : z4;
    }
    mark();

    store_split<SP, ROWS>(pA, hw, c32, sv);                     // aggregated messages -> planes
    __syncthreads();
    mark();

    f4 acc[1][1];
    issue(std::integral_constant<int, D>{});
    acc[0][0] = z4;
    mma_tile_split<SP, 4, 1, 1, ROWS, 256, 4, 0, true, TM_NODE_DEEP_PF>(pA, ring[0], acc, lane);             // W3
    {
        const f4 dh = fma4s(cnt, b3, acc[0][0]) / 30.0f;
        st4(tB + chunk_off(m, c4), hv + dh);
    }
    __syncthreads();
    mark();
    {   // LN1: fp32 in place (the FFN residual) + planes (the FFN input)
        float *p = tB + chunk_off(hw, c32);
        const f4 y = layer_norm_row(ld4(p), g1, be1);
        st4(p, y);
        store_split<SP, ROWS>(pB, hw, c32, y);
    }
    __syncthreads();
    mark();

    f4 out[1][1];
    out[0][0] = bout;
    static_for<0, 4>([&](auto C) {                              // FFN hidden 512 in four 128-wide chunks
        constexpr int c = decltype(C)::value;
        issue(std::integral_constant<int, 1 + 2 * c + D>{});
        acc[0][0] = bin[c];
        mma_tile_split<SP, 4, 1, 1, ROWS, 256, 4, 0, true, TM_NODE_DEEP_PF>(pB, ring[(1 + 2 * c) % NS], acc, lane);
        store_split<SP, ROWS>(pA, m, c4, gelu4(acc[0][0]));
        __syncthreads();
        mark();
        issue(std::integral_constant<int, 2 + 2 * c + D>{});
        mma_tile_split<SP, 4, 1, 1, ROWS, 256, 4, 0, true, TM_NODE_DEEP_PF>(pA, ring[(2 + 2 * c) % NS], out, lane);
        __syncthreads();
        mark();
    });
    {
        const int off = chunk_off(m, c4);
        st4(tA + off, ld4(tB + off) + out[0][0]);               // tA aliases pA: every wavefront is past its last read
    }
    __syncthreads();
    {   // LN2, mask, coalesced store; the new state goes into the planes pB for the projections
        f4 y = layer_norm_row(ld4(tA + chunk_off(hw, c32)), g2, be2);
        y = ok_h ? y * mk : z4;
        store_split<SP, ROWS>(pB, hw, c32, y);
        if (ok_h) st4(a.h_out + (size_t)grow * TM_H + 4 * c32, y);
    }
    mark();
    if constexpr (NPROJ > 0) {
        __syncthreads();
        static_for<0, 2 * NPROJ>([&](auto J) {
            constexpr int j = decltype(J)::value, k = j >> 1, half = j & 1;
            issue(std::integral_constant<int, 9 + j + D>{});
            acc[0][0] = half ? z4 : pb[k];
            mma_tile_split<SP, 4, 1, 1, ROWS, 256, 4, 0, true, TM_NODE_DEEP_PF>(pB, ring[(9 + j) % NS], acc, lane);
            if (ok_m) {
                float *dst = a.proj[pk[k]].P + (size_t)row_m * 256 + 128 * half + ncol;
                st4(dst, half && has_add[k] ? padd[k] + acc[0][0] : acc[0][0]);
            }
            mark();
        });
    }
}

// fragment image of one 128 x 128 block (see WImg in tmpnn_internal.h): [wv 8][c 4][plane 2][lane 64] x 16 B
__global__ void prep_wimg_kernel(const float *__restrict__ W, int ld, int n_rows, int k_valid, int k_wrap, char *__restrict__ dst) {
    const int idx = tm_bid() * tm_bdim() + tm_tid();      // (wv, c, lane)
    if (idx >= 8 * 4 * 64) return;
    const int lane = idx & 63, c = (idx >> 6) & 3, wv = idx >> 8, m = lane & 15, q = lane >> 4;
    const float *src = W + (size_t)(16 * wv + m) * ld + 32 * c + 8 * q;
    // blocks with fewer than 128 rows / fewer than 128 columns (k_valid, a multiple of 8): zero fragments beyond, nothing read —
    // except the k_wrap columns after k_valid, which come from the same row one `ld` back (load_wfrag_split)
    const int k = 32 * c + 8 * q;
    const bool ok = 16 * wv + m < n_rows && k < k_valid + k_wrap;
    if (k >= k_valid) src -= ld;
    const f4 z = f4{0.f, 0.f, 0.f, 0.f};
    const f4 v0 = ok ? ld4(src) : z, v1 = ok ? ld4(src + 4) : z;
    unsigned w4[4][2];
    SplitH2::split2(f2{v0.x, v0.y}, w4[0]);
    SplitH2::split2(f2{v0.z, v0.w}, w4[1]);
    SplitH2::split2(f2{v1.x, v1.y}, w4[2]);
    SplitH2::split2(f2{v1.z, v1.w}, w4[3]);
#pragma unroll
    for (int p = 0; p < 2; ++p)
        *reinterpret_cast<u4 *>(dst + (size_t)wv * 8192 + c * 2048 + p * 1024 + lane * 16) = u4{w4[0][p], w4[1][p], w4[2][p], w4[3][p]};
}
int launch_prep_wimg(const float *W, int ld, char *dst, hipStream_t st, int n_rows, int k_valid, int k_wrap) {
    prep_wimg_kernel<<<8, 256, 0, st>>>(W, ld, n_rows, k_valid, k_wrap, dst);
    return tm_check_launch("prep_wimg");
}

int launch_node_update_split(const NodeArgs &a, int64_t T, hipStream_t st) {
    // tile height (16..64 rows, one workgroup per CU) for load balance: every tile streams the same 0.8 MB of weights,
    // worth about `wcost` rows of (cheaper) matrix time
    const int64_t slots = tm_num_cus();
    const int wcost = 48, max_rows = 64;
    int best_rows = max_rows;
    int64_t best_cost = -1;
    for (int rows = max_rows; rows >= 16; rows -= 16) {
        const int64_t tiles = (T + rows - 1) / rows, rounds = (tiles + slots - 1) / slots;
        const int64_t cost = rounds * (rows + wcost);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_rows = rows; }
    }
    const int64_t tiles = (T + best_rows - 1) / best_rows;
    const int grid = (int)(tiles < slots ? tiles : slots);
    static const int deep = TM_DBG_INT("TMPNN_NODE_DEEP", 1);
    if (deep && a.img[0] && (T + 15) / 16 <= slots) {           // one 16-row tile per workgroup: the deep-prefetch form
        const int g16 = (int)((T + 15) / 16);
        const int np = (a.proj[0].P != nullptr) + (a.proj[1].P != nullptr);
        NodeArgs b = a;
        if (np == 1 && a.proj[0].P == nullptr) {                // compact: the present projection first
            b.proj[0] = a.proj[1];
            b.proj[1] = a.proj[0];
            b.img[9] = a.img[11];
            b.img[10] = a.img[12];
        }
#ifdef TMPNN_DEBUG_BUILD
        static const bool prof = TM_DBG_FLAG("TMPNN_NODE_PROF", false);
        if (prof && np == 2) {                                  // debug build: stage stamps of workgroup 0 (synchronises!)
            static unsigned long long *d_prof = nullptr;
            if (!d_prof) (void)hipMalloc(&d_prof, 32 * sizeof(unsigned long long));
            (void)hipMemsetAsync(d_prof, 0, 32 * sizeof(unsigned long long), st);
            node_update8_deep_kernel<2, TM_NODE_DEEP_D, true><<<g16, 512, 0, st>>>(b, d_prof);
            unsigned long long h[32];
            (void)hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "node_update8_deep stages (cycles since entry, wg 0): primed %llu | planes %llu | W3 %llu | LN1 %llu |", h[1] - h[0],
                    h[2] - h[0], h[3] - h[0], h[4] - h[0]);
            for (int k = 5; k < 13; ++k) fprintf(stderr, " %llu", h[k] - h[0]);
            fprintf(stderr, " | LN2 %llu | proj", h[13] - h[0]);
            for (int k = 14; k < 18; ++k) fprintf(stderr, " %llu", h[k] - h[0]);
            fprintf(stderr, "\n");
            return tm_check_launch("node_update8_deep");
        }
#endif
        if (np == 0) node_update8_deep_kernel<0, TM_NODE_DEEP_D><<<g16, 512, 0, st>>>(b);
        else if (np == 1) node_update8_deep_kernel<1, TM_NODE_DEEP_D><<<g16, 512, 0, st>>>(b);
        else node_update8_deep_kernel<2, TM_NODE_DEEP_D><<<g16, 512, 0, st>>>(b);
        return tm_check_launch("node_update8_deep");
    }
#ifdef TMPNN_DEBUG_BUILD
    static const bool prof4 = TM_DBG_FLAG("TMPNN_NODE_PROF", false);
    if (prof4 && a.img[0] && best_rows == 64) {                 // debug build: stage stamps of workgroup 0 (synchronises!)
        static unsigned long long *d_prof = nullptr;
        if (!d_prof) (void)hipMalloc(&d_prof, 32 * sizeof(unsigned long long));
        (void)hipMemsetAsync(d_prof, 0, 32 * sizeof(unsigned long long), st);
        node_update8_split_kernel<SplitH2, 4, true, true><<<grid, 512, 0, st>>>(a, d_prof);
        unsigned long long h[32];
        (void)hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "node_update8 (64 rows) stages (cycles since entry, wg 0):");
        for (int k = 1; k < 20 && h[k]; ++k) fprintf(stderr, " %llu", h[k] - h[0]);
        fprintf(stderr, "\n");
        return tm_check_launch("node_update8_split");
    }
#endif
#define TM_NODE8(NRB)                                                                    \
    if (a.img[0]) node_update8_split_kernel<SplitH2, NRB, true><<<grid, 512, 0, st>>>(a); \
    else node_update8_split_kernel<SplitH2, NRB, false><<<grid, 512, 0, st>>>(a)
    switch (best_rows) {
        case 16: TM_NODE8(1); break;
        case 32: TM_NODE8(2); break;
        case 48: TM_NODE8(3); break;
        default: TM_NODE8(4); break;
    }
#undef TM_NODE8
    return tm_check_launch("node_update8_split");
}
