// Node update (protein_mpnn_utils.py:823-825, 866-880 + the projections the next kernels gather), f16x2: tall-tile and one-tile forms.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "tmpnn_split.h"
#include "tmpnn_internal.h"
#include "tmpnn_head_body.h"

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
    // GEMM of the unit in `wf` with the NEXT unit's fragment requested meanwhile (u_next < 0: none). With images (round 6) its eight
    // requests ride one by one behind the GEMM's MFMA steps instead of standing in a row in front of it (~85 cycles of issue each).
    auto gemm = [&](const char *plane, f4 (&ac)[NRB][1], int u_next) {
        if constexpr (IMG) {
            const char *pn = a.img[u_next < 0 ? 0 : u_next] + (size_t)wv * 8192 + lane * 16;     // (none: unit 0 again, unconditional requests)
            constexpr int NS = 4 * NRB;
            mma_tile_split_ride<SP, 4, NRB, TM_NODE_PF, ROWS>(plane, wf, ac, lane, [&](auto S) {
                constexpr int s = decltype(S)::value;
                static_for<0, 8>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    if constexpr ((k * NS) / 8 == s) raw[k] = *reinterpret_cast<const f4 *>(pn + 2048 * (k >> 1) + 1024 * (k & 1));
                });
            });
        } else {
            if (u_next >= 0) issue(u_next);
            mma_tile_split<SP, 4, 1, NRB, ROWS, 256, 4, 0, true, TM_NODE_PF>(plane, wf, ac, lane);
        }
    };

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
            // the tile's per-row scalars (neighbour count, mask, the two sequence-table indices) are requested HERE, all four at once and
            // unconditionally (absent tables read a dummy), behind the rows and in front of their first use: round 5 found them as three
            // dependent load -> wait -> LDS-store groups of wavefront 0 in front of the tile's first barrier (one L2 round trip each, with
            // the other seven wavefronts waiting)
            const int rr = tid < ROWS ? tid : 0;                 // (ROWS = 48 is not a power of two)
            const bool okr = r0 + rr < a.T;
            const int gr = okr ? r0 + rr : r0;
            const bool ha0 = has0 && a.proj[0].add_tab != nullptr, ha1 = has1 && a.proj[1].add_tab != nullptr;
            const int32_t *ai0 = ha0 ? a.proj[0].add_idx : reinterpret_cast<const int32_t *>(a.cnt);
            const int32_t *ai1 = ha1 ? a.proj[1].add_idx : reinterpret_cast<const int32_t *>(a.cnt);
            float cv = 0.f, mv = 0.f;
            int i0 = 0, i1 = 0;
            if (tid < ROWS) {
                cv = a.cnt[gr];
                mv = a.mask[gr];
                i0 = ai0[gr];
                i1 = ai1[gr];
            }
#pragma unroll
            for (int it = 0; it < NRB; ++it) {
                const int row = 16 * it + (tid >> 5), c = tid & 31;
                const bool ok = r0 + row < a.T;
                const f4 z4 = f4{0.f, 0.f, 0.f, 0.f};
                store_split<SP, ROWS>(pA, row, c, ok ? v[it] : z4);
                st4(tB + chunk_off(row, c), ok ? hvv[it] : z4);
            }
            mark();
            if (tid < ROWS) {
                s_cnt[tid] = okr ? cv : 0.f;
                s_mask[tid] = okr ? mv : 0.f;
                if (ha0) s_aidx[0][tid] = okr ? i0 : 0;
                if (ha1) s_aidx[1][tid] = okr ? i1 : 0;
            }
        }
        __syncthreads();
        mark();

        f4 acc[NRB][1];
        split_raw();
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = f4{0.f, 0.f, 0.f, 0.f};
        gemm(pA, acc, 1);
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
            {
                const f4 b = ld4(s_par + P_BIN + 128 * c + ncol);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b;
            }
            gemm(pB, acc, 2 + 2 * c);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) store_split<SP, ROWS>(pA, 16 * rb + m, c4, gelu4(acc[rb][0]));
            __syncthreads();
            mark();
            split_raw();                        // W_out chunk c
            gemm(pA, out, c < 3 ? 3 + 2 * c : (has0 || has1) ? first_proj : tile + (int)tm_nblk() < n_tiles ? 0 : -1);
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
                {
                    const f4 b = half ? f4{0.f, 0.f, 0.f, 0.f} : ld4(s_par + P_BA + 128 * k + ncol);
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = b;
                }
                gemm(pB, acc, !half ? 10 + 2 * k : (k == 0 && has1) ? 11 : tile + (int)tm_nblk() < n_tiles ? 0 : -1);
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
template <int NPROJ, int D, bool PROF = false>
__device__ __forceinline__ void node_deep_body(const NodeArgs &a, unsigned long long *prof) {
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
    // round 6: behind the prologue the fragment requests ride two by two behind the four MFMA steps of the unit being multiplied
    // (eight global_loads in a row cost their wavefront ~85 cycles of issue each — as long as the unit's MFMAs)
    auto issue_piece = [&](auto P, auto K) {
        constexpr int p = decltype(P)::value, k = decltype(K)::value;
        if constexpr (p < NPOS) {
            const char *src = a.img[unit_at(p)] + (size_t)wv * 8192 + lane * 16;
            ring[p % NS][0][k >> 1].p[k & 1] = *reinterpret_cast<const u4 *>(src + 2048 * (k >> 1) + 1024 * (k & 1));
        }
    };
    auto gemm = [&](const char *plane, auto PU, auto PI, f4 (&ac)[1][1]) {      // unit PU from its ring slot; the pieces of position PI ride
        mma_tile_split_ride<SP, 4, 1, TM_NODE_DEEP_PF, ROWS>(plane, ring[decltype(PU)::value % NS], ac, lane, [&](auto S) {
            constexpr int s = decltype(S)::value;
            issue_piece(PI, std::integral_constant<int, 2 * s>{});
            issue_piece(PI, std::integral_constant<int, 2 * s + 1>{});
        });
    };
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
    acc[0][0] = z4;
    gemm(pA, std::integral_constant<int, 0>{}, std::integral_constant<int, D>{}, acc);                       // W3
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
        acc[0][0] = bin[c];
        gemm(pB, std::integral_constant<int, 1 + 2 * c>{}, std::integral_constant<int, 1 + 2 * c + D>{}, acc);
        store_split<SP, ROWS>(pA, m, c4, gelu4(acc[0][0]));
        __syncthreads();
        mark();
        gemm(pA, std::integral_constant<int, 2 + 2 * c>{}, std::integral_constant<int, 2 + 2 * c + D>{}, out);
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
            acc[0][0] = half ? z4 : pb[k];
            gemm(pB, std::integral_constant<int, 9 + j>{}, std::integral_constant<int, 9 + j + D>{}, acc);
            if (ok_m) {
                float *dst = a.proj[pk[k]].P + (size_t)row_m * 256 + 128 * half + ncol;
                st4(dst, half && has_add[k] ? padd[k] + acc[0][0] : acc[0][0]);
            }
            mark();
        });
    }
}

template <int NPROJ, int D, bool PROF = false>
__global__ __launch_bounds__(512) void node_update8_deep_kernel(NodeArgs a, unsigned long long *prof = nullptr) {
    node_deep_body<NPROJ, D, PROF>(a, prof);
}

// Small launches, last decoder layer (round 6): the node update of a workgroup's 16 residues and the ddG head of the SAME 16 residues in one
// launch — the head needs the new state of exactly these rows, the previous decoder state and the sequence embedding: no grid-wide
// dependency, and a launch costs 2.5 us of dispatch + 2-3 us of start-up whatever it does (tools/gap_probe.py). The new state goes to
// global memory as before (h_out is an output of the forward); the barrier between the two bodies makes the workgroup's own rows
// visible to all of its threads (stores complete at L2 before it, the rows were never in this CU's L1). Bodies unchanged: bit-identical
// to the two launches (tools/dbg_fused.py).
__global__ __launch_bounds__(512) void node_head_fused_kernel(NodeArgs a, HeadArgs h) {
    node_deep_body<0, TM_NODE_DEEP_D>(a, nullptr);
    __syncthreads();
    head8_body<SplitH2, 1, true>(h);
}

int launch_node_update_split(const NodeArgs &a, int64_t T, hipStream_t st, const HeadArgs *head, bool *head_ran) {
    if (head_ran) *head_ran = false;
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
        if (np == 0 && head && head->img[0]) {                  // + the ddG head of the same rows (the caller checked node_head_fusable)
            node_head_fused_kernel<<<g16, 512, 0, st>>>(b, *head);
            if (head_ran) *head_ran = true;
            return tm_check_launch("node_head_fused");
        }
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

// the fused launch exists for f16x2 handles with fragment images, when every workgroup has one 16-row tile (the deep form's condition)
bool node_head_fusable(int mode, int64_t T) {
    static const int deep = TM_DBG_INT("TMPNN_NODE_DEEP", 1);
    static const bool split = TM_DBG_FLAG("TMPNN_NODE_SPLIT", true) && TM_DBG_FLAG("TMPNN_HEAD_SPLIT", true);   // (the debug library's fp32-form switches)
    return mode == TM_MM_F16X2 && deep && split && T > 0 && (T + 15) / 16 <= (int64_t)tm_num_cus();
}
