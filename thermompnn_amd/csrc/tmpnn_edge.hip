// Edge update of an encoder layer (protein_mpnn_utils.py:826-839), split-precision forms: f16x2 = enc_edge8_rp_kernel, bf16x3 = enc_edge8_split_kernel.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "tmpnn_split.h"
#include "tmpnn_internal.h"

// ------------------------------------------------------------------------------------------------
// enc_edge, split-precision form (8 wavefronts, 1 workgroup per CU): same pipeline as enc_edge8_kernel
// (tmpnn_layers.hip) with the three 128x128 GEMMs on the 16-bit matrix cores. GEMM inputs live in LDS as plane tiles;
// the LayerNorm input is an fp32 tile aliased on the x planes. The next residue's fp32 tile lands in an LDS staging
// buffer by LDS-DMA under GEMM 1 and is split into the e planes during the LayerNorm/store phase. Residual: bf16x3
// re-joins the e planes (exact); f16x2 keeps the fp32 tile (two staging buffers, alternating).
// ------------------------------------------------------------------------------------------------
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
    char *const tY = tE;
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
    unsigned long long c_begin = 0, w_begin = 0;
    if (PROF) { c_begin = __builtin_readcyclecounter(); w_begin = wall_clock64(); }
    for (; i < tr.end; i += tr.step) {
        float *tile_g = a.hE + (size_t)i * TM_KS * TM_H;
        const int inext = i + tr.step;
        const bool has_next = inext < tr.end;
        const int ipf = has_next ? inext : i;              // prefetch target (the last iteration re-reads its own tile)
        const int ipf2 = ipf + tr.step < tr.end ? ipf + tr.step : ipf;
        const int nidx_pub = nidx;                         // list of tile ipf, requested during the previous iteration
        // the next tile's requests ride behind the MFMAs of GEMM 1, one per step (round 6: four global_loads in a row in front of the
        // GEMM cost the wavefront ~85 cycles of issue each)
        const float *src = a.hE + (size_t)ipf * TM_KS * TM_H;             // wave-uniform base + per-thread offset
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = gai + gcj[rb];
        mma_tile_split_ride<SP, 4, 3, TM_EDGE_PF>(tE, w11, acc, lane, [&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s == 1) { if (tid < TM_TILE) nidx = a.E_idx[(size_t)ipf2 * TM_KS + tid]; }
#if TM_ABL_NOLOAD
            if constexpr (s % 3 == 0 && s >= 3) e_nxt[s / 3 - 1] = e_cur[s / 3 - 1];
#else
            if constexpr (s % 3 == 0 && s >= 3) e_nxt[s / 3 - 1] = ld4(src + (eoff + 16 * (s / 3 - 1) * TM_H));
#endif
        });
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

#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b12;
        // ... and its gathered node terms behind the MFMAs of GEMM 2 (the list they need was published in front of the barrier above)
        mma_tile_split_ride<SP, 4, 3, TM_EDGE_PF>(tX, w12, acc, lane, [&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s == 1) gai = ld4(a.P + (size_t)ipf * 256 + ucol);
            if constexpr (s % 3 == 0 && s >= 3) gcj[s / 3 - 1] = ld4(prow_of(s_idx[cur ^ 1][16 * (s / 3 - 1) + m], ipf));
        });
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
    if (PROF && tm_bid() == 0 && tm_tid() == TM_PROF_TID) {               // shader cycles and 100 MHz ticks of the loop: the clock under THIS load
        prof[11] = __builtin_readcyclecounter() - c_begin;
        prof[12] = wall_clock64() - w_begin;
    }
}

int launch_enc_edge_split(int mode, const EncW &e, const float *P, float *hE, const int32_t *E_idx, int64_t T, hipStream_t st) {
    const bool h2 = mode == TM_MM_F16X2;
    EdgeArgsB a{e.W11 + 128, e.W12, e.b12, e.W13, e.b13, e.norm3_w, e.norm3_b, P, hE, E_idx, (int)T,
                h2 ? tm_find_wimg(e.W11 + 128) : nullptr, h2 ? tm_find_wimg(e.W12) : nullptr, h2 ? tm_find_wimg(e.W13) : nullptr,
                h2 ? tm_find_wimgp(e.W11 + 128) : nullptr, h2 ? tm_find_wimgp(e.W12) : nullptr, h2 ? tm_find_wimgp(e.W13) : nullptr};
    // f16x2, large launches: one wavefront per 16-row block, one wavefront per SIMD (tmpnn_edge_wave.hip)
    if (h2 && a.imgp11 && a.imgp12 && a.imgp13 && enc_edge_wave_wanted(T)) return launch_enc_edge_wave(a, T, st);
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
            const long long tiles0 = T >= 8 * cap ? (T / 8 + cap / 8 - 1) / (cap / 8) : (T + cap - 1) / cap;       // tiles of workgroup 0 (xcd_tile_range)
            fprintf(stderr, "enc_edge phases (cycles, wg 0): gemm1 %llu gelu+split %llu bar %llu gather+gemm2 %llu gelu+split %llu bar %llu gemm3 %llu resid+stats %llu bar %llu split+ln+store %llu bar %llu; loop %llu cycles in %llu ticks of 100 MHz = %.3f GHz, %lld tiles = %.0f cycles per tile\n",
                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[12] ? h[11] / (h[12] * 10.0) : 0.0,
                    tiles0, tiles0 ? (double)h[11] / tiles0 : 0.0);
#endif
        } else if (T < ((int64_t)1 << 22)) {
            enc_edge8_rp_kernel<SplitH2, false, true><<<grid, 512, 0, st>>>(a);      // projection table < 4 GB: 32-bit gather offsets
        } else {
            enc_edge8_rp_kernel<SplitH2><<<grid, 512, 0, st>>>(a);
        }
    }
    return tm_check_launch("enc_edge_split");
}
