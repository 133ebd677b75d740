// Edge update of an encoder layer (protein_mpnn_utils.py:826-839), split-precision form: enc_edge8_rp_kernel (f16x2 and bf16x3).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "tmpnn_split.h"
#include "tmpnn_internal.h"

// ------------------------------------------------------------------------------------------------
// enc_edge, register-prefetch form (both split precisions since round 6; bf16x3 with its own budget, see SP::EXACT below — until
// then it staged the tile through LDS-DMA + an fp32 LDS tile, 139 KB of LDS and 0.58 ms per launch of the bench batch): the next residue's fp32 tile is
// loaded straight into the accumulator layout (row 16 rb + m, columns 16 wv + 4 q: one 16-byte load per row block)
// at the top of the iteration, split into the e planes after GEMM 3 and kept in registers as the fp32 residual of the
// next iteration. No LDS staging, no LDS-DMA (whose conservative vmcnt(0) waits serialised the store phase), biases and
// LayerNorm parameters live in registers, every global access of the loop is unconditional.
// ------------------------------------------------------------------------------------------------
// OFF32: see msg8_rp_kernel (32-bit gather offsets when the projection table is smaller than 4 GB).
template <typename SP, bool PROF = false, bool OFF32 = false>
__global__ __launch_bounds__(512, 2) void enc_edge8_rp_kernel(EdgeArgsB a, unsigned long long *prof = nullptr) {
    // TMPNN_EDGE_PROF=1 (debug library): phase timing of one wavefront of workgroup 0 in scalar registers (s_memtime + SALU adds, written out once
    // behind the loop; round 6 — a timer that does a global read-modify-write per mark waits for every request in flight at every mark: +15 %)
    unsigned t_last = 0, t_acc[11] = {};
    auto mark = [&](int k) {
        if constexpr (PROF) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            if (k >= 0) t_acc[k] += t - t_last;
            t_last = t;
        }
    };
    constexpr int TILEB = SP::NP * SPLIT_PLANE_BYTES;
    __shared__ __attribute__((aligned(16))) char tE[TILEB];
    __shared__ __attribute__((aligned(16))) char tX[TILEB];              // x planes; later the fp32 LayerNorm input
    // GEMM 2's output planes live where the e planes were: GEMM 1 was their last reader (every wavefront is past the barrier behind
    // it), the next tile's e planes are written only behind the barrier that follows GEMM 3. Two plane tiles instead of three:
    // 53 KB of LDS, every LDS offset below 64 KB (an offset above costs an address VGPR + a v_or each: 10 VALU per tile).
    // bf16x3 (round 6): its three planes re-join exactly, so the e planes ARE the residual (no fp32 copy of the tile in registers) and GEMM 2's
    // output gets a tile of its own (3 x 36 KB of LDS); the per-column parameters come from LDS. 48 weight VGPRs per matrix leave no room otherwise.
    // Round 6, THREE barriers per tile instead of four: GEMM 2's output planes and the fp32 LayerNorm tile have tiles of their own (98 KB of LDS
    // in f16x2, 141 KB in bf16x3; until then tY aliased the e planes and tO the x planes), the next tile's e planes are written in the residual
    // phase (GEMM 1 read the old ones two barriers ago), so nothing separates a tile's LayerNorm / store phase from the next tile's GEMM 1 and
    // GELU 1: the older wavefront of a SIMD runs ahead into the matrix phase while the younger finishes its rows, instead of waiting for it.
    __shared__ __attribute__((aligned(16))) char tY[TILEB];
    __shared__ __attribute__((aligned(16))) float tO[TM_TILE * TM_H];
    __shared__ __attribute__((aligned(16))) float s_par[SP::EXACT ? 4 : 1][TM_H];
    __shared__ __attribute__((aligned(16))) float s_stat[TM_TILE][TM_STAT_LD];
    __shared__ int s_idx[2][TM_TILE];
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
    constexpr int EPF = SP::EXACT ? 1 : TM_EDGE_PF;                      // B-fragment prefetch distance (three planes per fragment in bf16x3)
    const f4 z4p = f4{0.f, 0.f, 0.f, 0.f};
    f4 b12r = z4p, b13r = z4p, g4r = z4p, be4r = z4p;                    // (bf16x3 reads these four from LDS instead: s_par)
    if constexpr (SP::EXACT) {
        if (tid < 128) st4(&s_par[tid >> 5][4 * (tid & 31)], ld4((tid < 32 ? a.b12 : tid < 64 ? a.b13 : tid < 96 ? a.g3 : a.be3) + 4 * (tid & 31)));
    } else {
        b12r = ld4(a.b12 + ncol); b13r = ld4(a.b13 + ncol);
        g4r = ld4(a.g3 + 4 * c32); be4r = ld4(a.be3 + 4 * c32);
    }
    auto par = [&](int k, int col, const f4 &reg) -> f4 { if constexpr (SP::EXACT) return ld4(&s_par[k][col]); else return reg; };

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
        mma_tile_split_ride<SP, 4, 3, EPF>(tE, w11, acc, lane, [&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s == 1) { if (tid < TM_TILE) nidx = a.E_idx[(size_t)ipf2 * TM_KS + tid]; }
#if TM_ABL_NOLOAD
            if constexpr (s % 3 == 0 && s >= 3) e_nxt[s / 3 - 1] = e_cur[s / 3 - 1];
#else
            if constexpr (!SP::EXACT && s % 3 == 0 && s >= 3) e_nxt[s / 3 - 1] = ld4(src + (eoff + 16 * (s / 3 - 1) * TM_H));
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
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = par(0, ncol, b12r);
        // ... and its gathered node terms behind the MFMAs of GEMM 2 (the list they need was published in front of the barrier above)
        // (bf16x3, 48 weight VGPRs per matrix: every request one phase later — the e rows behind GEMM 2, the gathers behind GEMM 3's end —
        //  so that the 28 registers they land in are not all live through the three GEMMs)
        mma_tile_split_ride<SP, 4, 3, EPF>(tX, w12, acc, lane, [&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (SP::EXACT) {
                if constexpr (s % 3 == 0 && s >= 3) e_nxt[s / 3 - 1] = ld4(src + (eoff + 16 * (s / 3 - 1) * TM_H));
            } else {
                if constexpr (s == 1) gai = ld4(a.P + (size_t)ipf * 256 + ucol);
                if constexpr (s % 3 == 0 && s >= 3) gcj[s / 3 - 1] = ld4(prow_of(s_idx[cur ^ 1][16 * (s / 3 - 1) + m], ipf));
            }
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
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = par(1, ncol, b13r);
        mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, EPF>(tY, w13, acc, lane);
        if constexpr (SP::EXACT) {
            gai = ld4(a.P + (size_t)ipf * 256 + ucol);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) gcj[rb] = ld4(prow_of(s_idx[cur ^ 1][16 * rb + m], ipf));
        }
        mark(6);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            f4 v;                                                            // residual: the fp32 tile, or (bf16x3) the exact re-join of its planes
            if constexpr (SP::EXACT) v = load_joined<SP>(tE, 16 * rb + m, c4) + acc[rb][0];
            else v = e_cur[rb] + acc[rb][0];
            st4(tO + chunk_off(16 * rb + m, c4), v);
#if TM_ABL_NOLN
            (void)q;
#else
            row_stats_partial16(v, &s_stat[16 * rb + m][2 * wv], q);
#endif
            store_split<SP>(tE, 16 * rb + m, c4, e_nxt[rb]);                 // the next tile's e planes (this thread's own slot: the residual above read it)
            e_cur[rb] = e_nxt[rb];
        }
        bool okr[3];                                                         // rows of this thread's LayerNorm phase that have a neighbour (s_idx[cur] is
#pragma unroll                                                               //  rewritten by the next tile's GELU-1 phase, which no barrier separates from it)
        for (int it = 0; it < 3; ++it) okr[it] = s_idx[cur][6 * wv + 2 * it + (lane >> 5)] >= 0;
        mark(7);
        __syncthreads();                                                     // tO + stats + the next e planes complete
        mark(8);

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
            const f4 g4 = par(2, 4 * c32, g4r), be4 = par(3, 4 * c32, be4r);
            const f2 s01 = f2{g4.x, g4.y} * rstd, s23 = f2{g4.z, g4.w} * rstd;
            const f2 t01 = __builtin_elementwise_fma(f2{-mean, -mean}, s01, f2{be4.x, be4.y});
            const f2 t23 = __builtin_elementwise_fma(f2{-mean, -mean}, s23, f2{be4.z, be4.w});
            const f2 y01 = __builtin_elementwise_fma(f2{x4.x, x4.y}, s01, t01), y23 = __builtin_elementwise_fma(f2{x4.z, x4.w}, s23, t23);
            const f4 y = f4{y01.x, y01.y, y23.x, y23.y};
            // rows without a neighbour keep the zeros the featurizer wrote: store zeros again (no divergent branch)
            st4(tile_g + (soff + 2 * it * TM_H), okr[it] ? y : f4{0.f, 0.f, 0.f, 0.f});
        }
        cur ^= 1;
        mark(9);
        mark(10);                                                            // (no barrier here any more)
    }
    if (PROF && tm_bid() == 0 && tm_tid() == TM_PROF_TID) {               // shader cycles and 100 MHz ticks of the loop: the clock under THIS load
#pragma unroll
        for (int k = 0; k < 11; ++k) prof[k] = t_acc[k];
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
    if (mode == TM_MM_BF16X3) {
        if (T < ((int64_t)1 << 22)) enc_edge8_rp_kernel<SplitBF3, false, true><<<grid, 512, 0, st>>>(a);
        else enc_edge8_rp_kernel<SplitBF3><<<grid, 512, 0, st>>>(a);
    }
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
