// Message pass of an encoder / decoder layer (protein_mpnn_utils.py:816-823, 859-866), split-precision forms (f16x2 and bf16x3):
// msg8_rp_kernel.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "tmpnn_split.h"
#include "tmpnn_internal.h"

// ------------------------------------------------------------------------------------------------
// message kernel, split-precision forms (8 wavefronts, 1 workgroup per CU). Same arithmetic as msg_kernel (tmpnn_layers.hip):
// Ssum_i = sum_k ma_ik gelu(W2 gelu(pre_ik) + b2).
// ------------------------------------------------------------------------------------------------
// Register-prefetch form: the next residue's fp32 tile is loaded in row layout at the top of the iteration and split into the
// e planes once GEMM 1 has consumed the current ones. Both split precisions run it: f16x2 (168-172 VGPRs) and bf16x3 (three planes,
// 48 weight VGPRs per matrix: 228-230 VGPRs, no scratch; until round 5 bf16x3 staged the tile through LDS-DMA + an fp32 LDS tile
// and summed over K through LDS: 0.32 ms per launch of the bench batch against 0.284 ms in this form).
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

    constexpr bool PERM = SP::NP == 2;                          // f16x2: the message pass's K order (perm_c4, tmpnn_split.h)
    WFragS<SP> w1[1][4], w2[1][4];
    load_wfrag_auto<SP, PERM>(PERM ? a.imgp1 : a.img1, a.W1e, a.ld1, wv, lane, w1[0]);
    load_wfrag_auto<SP, PERM>(PERM ? a.imgp2 : a.img2, a.W2, TM_H, wv, lane, w2[0]);
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const int c4s = PERM ? perm_c4(c4) : c4;                    // where this thread's column group goes in a plane row
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
    const int pcs = PERM ? perm_c4(pc) : pc;
    const unsigned eoff = (unsigned)(prow * TM_H + 4 * pc);         // this thread's offset inside any e tile
    auto fetch_into = [&](f4 (&dst)[3], int ii) {
        const float *src = a.hE + (size_t)__builtin_amdgcn_readfirstlane(ii) * (TM_KS * TM_H);      // wave-uniform: scalar base + lane offset
#pragma unroll
        for (int it = 0; it < 3; ++it) dst[it] = ld4(src + (eoff + 2 * it * TM_H));
    };
    auto fetch_tile = [&](int ii) { fetch_into(e_nxt, ii); };
    auto split_tile = [&]() {
#pragma unroll
        for (int it = 0; it < 3; ++it) store_split<SP>(tE, prow + 2 * it, pcs, e_nxt[it]);
    };

    TileRange tr = xcd_tile_range(a.T);
    tr.begin += a.i0;                                           // residues [i0, i0 + T) of the packed axis
    tr.end += a.i0;
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
            store_split<SP>(tA, 16 * rb + m, c4s, gelu4(v));
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
        // Round 6: the seven requests ride one by one behind the MFMA steps of GEMM 2, in that order (in a row in front of it they
        // cost the wavefront ~85 cycles of issue each).
        const int ipf3 = ipf + tr.step < tr.end ? ipf + tr.step : ipf;
        const float *src3 = a.hE + (size_t)__builtin_amdgcn_readfirstlane(ipf3) * (TM_KS * TM_H);
        mark(3);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = bias2;
        mma_tile_split_ride<SP, 4, 3, TM_MSG_PF>(tA, w2, acc, lane, [&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s == 0) g0 = ld4(a.P + (size_t)__builtin_amdgcn_readfirstlane(ipf) * 256 + ucol);
            if constexpr (s >= 1 && s <= 3) {
                const int j0 = s_idx[cur ^ 1][16 * (s - 1) + m];
                const int j = j0 < 0 ? ipf : j0;
                if constexpr (OFF32) gj[s - 1] = ld4(a.P + ((unsigned)j * 256u + (128u + ucol)));
                else gj[s - 1] = ld4(a.P + (size_t)j * 256 + 128 + ncol);
            }
#if !TM_ABL_NOLOAD
            if constexpr (s >= 5 && s <= 9 && (s & 1)) e_nxt[(s - 5) >> 1] = ld4(src3 + (eoff + 2 * ((s - 5) >> 1) * TM_H));
#endif
        });
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

// ------------------------------------------------------------------------------------------------
// One wavefront per residue, no workgroup barrier in the loop (round 5; f16x2, launches with >= TM_MSG_WAVE_MIN residues per wavefront).
// The 8-wavefront form above splits the 128 output columns of a GEMM over the wavefronts of a workgroup, so the next GEMM (whose K axis
// IS those columns) starts behind a barrier, every epilogue ends in one, and the SIMD's two wavefronts run matrix and vector phases in
// lock-step: pipe times ADD (docs/NOTEBOOK.md 9.2-9.5). Here a wavefront owns 16 edge rows for the whole chain
//     e (global) -> B operand -> GEMM 1 -> GELU -> B operand -> GEMM 2 -> GELU -> masked sum over the rows
// and the activations never leave its registers: v_mfma_f32_16x16x32_f16 with A = weights, B = activations leaves lane (n, q) holding
// output columns 16 cb + 4 q + {0..3} of edge row n — and the 8 K values lane (n, q) feeds into step c of the NEXT GEMM may be ANY 8, as
// long as the weight fragment of that step uses the same assignment: the K-permuted fragment images (prep_wimg_kernel, perm) pair
// accumulator blocks 2 c and 2 c + 1. The same order makes the loads of e 64 contiguous bytes per four lanes. Weights: both matrices'
// images (2 x 64 KB) in LDS, read as conflict-free 1 KB fragments; the LDS read volume per residue equals that of the A-fragment
// reads of the 8-wavefront form. The wavefronts of a workgroup share nothing but those images, drift apart, and one's matrix phase
// runs beside the other's vector phase.
// ------------------------------------------------------------------------------------------------
// Work unit = one residue of one wavefront (~25 us), so this form takes whole multiples of 8 residues per workgroup; the launcher hands
// it floor(T / (8 #CUs)) x 8 #CUs residues and the remainder (< 8 per workgroup) to the 8-wavefront form — which computes the same bits
// (perm_c4, tmpnn_split.h). TM_MSG_WAVE_MIN: the smallest multiple worth a second launch and a 128 KB LDS fill per workgroup.
#define TM_MSG_WAVE_MIN 2
#define TM_MSG_WAVE_ILV 2       // units (accumulators) per fragment request group, their partial products interleaved term by term

// acc[cb] += W[16 cb .. 16 cb + 16, :] . x over K = 128: 32 (step, block) units of 3 MFMAs; wl = the image in LDS + 16 * lane.
// Fragments double-buffered by group of ILV units. A GEMM of 96 MFMAs takes 3 700 cycles here (38 per MFMA against 16 of pipe time)
// WHATEVER the request depth (1 / 3 / 5 units ahead) or the interleave (1 / 2 / 4 accumulators): it is neither LDS latency nor the
// dependence of the three products on one accumulator — the chip runs these kernels at its power limit (docs/NOTEBOOK.md 9.10).
// ride(G), G = 0..15: what the caller wants issued behind the six MFMAs of group G (round 6: the next block's global requests, one per
// group — a global_load costs its wavefront ~85 cycles of issue when 17 of them stand in a row, nothing behind a group of MFMAs).
template <typename R>
__device__ __forceinline__ void mma_wave_lds(const char *wl, const u4 (&x)[4][2], f4 (&acc)[8], R &&ride) {
    constexpr int NS = 32, ILV = TM_MSG_WAVE_ILV, NG = NS / ILV;
    u4 w[2][ILV][2];
#if TM_ABL_NOMFMA
    static_for<0, NG>([&](auto G) { ride(G); });
    return;
#endif
    auto request = [&](int g, int half) {
#if defined(TM_ABL_WAVE_NOLDS)
        if (g > 1) return;                        // timing ablation (debug builds): the first two groups' fragments, reused
#endif
#pragma unroll
        for (int k = 0; k < ILV; ++k) {
            const int s = g * ILV + k;
            w[half][k][0] = *reinterpret_cast<const u4 *>(wl + (s & 7) * 8192 + (s >> 3) * 2048);
            w[half][k][1] = *reinterpret_cast<const u4 *>(wl + (s & 7) * 8192 + (s >> 3) * 2048 + 1024);
        }
    };
    request(0, 0);
#define TM_HF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0)
    static_for<0, NG>([&](auto G) {
        constexpr int g = decltype(G)::value, h = g & 1;
        if (g + 1 < NG) request(g + 1, h ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const u4 (&xs)[2] = x[(g * ILV) >> 3];
#pragma unroll
        for (int k = 0; k < ILV; ++k) acc[(g * ILV + k) & 7] = TM_HF(w[h][k][1], xs[0], acc[(g * ILV + k) & 7]);      // l h
#pragma unroll
        for (int k = 0; k < ILV; ++k) acc[(g * ILV + k) & 7] = TM_HF(w[h][k][0], xs[1], acc[(g * ILV + k) & 7]);      // h l
#pragma unroll
        for (int k = 0; k < ILV; ++k) acc[(g * ILV + k) & 7] = TM_HF(w[h][k][0], xs[0], acc[(g * ILV + k) & 7]);      // h h
        __builtin_amdgcn_sched_barrier(0);
        ride(G);
        __builtin_amdgcn_sched_barrier(0);
    });
#undef TM_HF
}
template <bool DEC, bool OFF32, bool PROF = false>
__global__ __launch_bounds__(512, 2) void msg8_wave_kernel(MsgArgsB a, unsigned long long *prof = nullptr) {
    // TMPNN_MSG_PROF=1 (debug library): phase timing of one wavefront of workgroup 0 in scalar registers (s_memtime + SALU adds, written out
    // once behind the loop — a timer that does a global read-modify-write per mark waits for every request in flight at every mark)
    unsigned t_last = 0, t_acc[8] = {};
    auto mark = [&](int k) {
#ifndef TM_MSG_PROF_NOMARKS
#define TM_MSG_PROF_NOMARKS 0                               // 1: only the loop totals of the wavefronts, the code between them as shipped
#endif
        if constexpr (PROF && !TM_MSG_PROF_NOMARKS) {
            __builtin_amdgcn_sched_barrier(0);              // (s_memtime is no scheduling barrier by itself: the phases would smear)
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            if (k >= 0) t_acc[k] += t - t_last;
            t_last = t;
        }
    };
    __shared__ __attribute__((aligned(16))) char sW[2 * TM_WIMG_BYTES];
    __shared__ __attribute__((aligned(16))) float s_g0[8][TM_H];    // the residue's own projection row, per wavefront
    __shared__ __attribute__((aligned(16))) float s_b2[TM_H];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, n = lane & 15, q = lane >> 4;
    for (int o = tid * 16; o < TM_WIMG_BYTES; o += 512 * 16) {
        *reinterpret_cast<u4 *>(sW + o) = *reinterpret_cast<const u4 *>(a.imgp1 + o);
        *reinterpret_cast<u4 *>(sW + TM_WIMG_BYTES + o) = *reinterpret_cast<const u4 *>(a.imgp2 + o);
    }
    if (tid < TM_H / 4) st4(&s_b2[4 * tid], ld4(a.b2 + 4 * tid));
    __syncthreads();                                            // the only barrier of the kernel
    const char *wl1 = sW + 16 * lane, *wl2 = sW + TM_WIMG_BYTES + 16 * lane;
    float *g0s = s_g0[wv];

    const TileRange tr = xcd_tile_range(a.T);
    // Residues of this workgroup: k = 0 .. n_wg - 1 <-> i = tr.begin + k tr.step; SIMD s = wavefronts s and s + 4 serves k = s + 4 t,
    // t < m. Round 6: the two do NOT take m / 2 each. They do not share the SIMD evenly — the first-dispatched one (0..3) gets the matrix
    // and vector slots it asks for at 29-30 k cycles a residue whatever its partner does, the partner (4..7) the rest: 60 k a residue beside
    // it, 22 k alone. With equal shares wavefronts 0..3 were through after 234 k cycles and 4..7 after 334 k, alone for the last 30 % of
    // the launch; with 11 of 16 against 5 both end at 313-316 k (TMPNN_MSG_PROF, "wave loops"; docs/NOTEBOOK.md 10.3f). An LDS ticket
    // counter (every wavefront draws its next residue) ended as level in workgroup 0 but was 3 % slower over the launch than 11 : 5 — its
    // granule is a residue of the slow wavefront, 60 k cycles. Any wavefront computes the same bits for a residue.
#ifndef TM_MSG_WAVE_OLD_16TH
#define TM_MSG_WAVE_OLD_16TH 11      // sixteenths of a SIMD's residues that its first-dispatched wavefront takes
#endif
    const int n_wg = tr.begin < tr.end ? (tr.end - tr.begin + tr.step - 1) / tr.step : 0;
    const int wvs = tm_wave(tid), simd = wvs & 3;
    const int m_simd = (n_wg - simd + 3) / 4;                     // residues of this SIMD (n_wg >= 0)
    const int n_old = (m_simd * TM_MSG_WAVE_OLD_16TH + 8) / 16;
    const int t0 = wvs < 4 ? 0 : n_old, nt = wvs < 4 ? n_old : m_simd - n_old;      // this wavefront: t = t0 .. t0 + nt - 1
    auto res_at = [&](int t) { return __builtin_amdgcn_readfirstlane(tr.begin + (simd + 4 * (t0 + t)) * tr.step); };
    int t_res = 0;
    // (the eight wavefronts of a workgroup are tr.step = 32 residues apart; giving them eight CONSECUTIVE residues, whose neighbour lists
    //  overlap, so that their gathers meet in the CU's L1 was measured in round 6: nil — 0.4182 / 0.4193 / 0.4175 against 0.4189 / 0.4194 /
    //  0.4186 of the featurizer's time in the same run)
    if (nt <= 0) return;
    int i = res_at(0);
    const unsigned uq = 4u * (unsigned)q;
    const unsigned eoff = (unsigned)(n * TM_H) + uq;           // this lane's offset inside a 16-row block of e

    auto idx_of = [&](int ii, int bb) { return (a.E_idx + ((size_t)__builtin_amdgcn_readfirstlane(ii) * TM_KS + 16 * __builtin_amdgcn_readfirstlane(bb)))[(unsigned)n]; };
    f4 e_n[8], g_n[8];
    float mk_n = 1.f;
    // operands of a block as 16 pieces: piece G rides behind MFMA group G of GEMM 1 (pieces 0..7: the gathered projection row, 16
    // bytes per lane each; 8..15: the e rows; the mask gather goes with piece 8). Addresses are formed ONCE per block (BlockAddr).
    struct BlockAddr { unsigned goff; const float *pj; const float *src; unsigned jj; };
    auto block_addr = [&](int ii, int bb, int j) {
        const int jj = j < 0 ? ii : j;
        BlockAddr r;
        r.jj = (unsigned)jj;
        r.goff = (unsigned)jj * 256u + (128u + uq);
        r.pj = a.P + (size_t)jj * 256 + 128 + uq;
        r.src = a.hE + ((size_t)__builtin_amdgcn_readfirstlane(ii) * TM_KS + 16 * __builtin_amdgcn_readfirstlane(bb)) * TM_H;
        return r;
    };
    auto issue_piece = [&](const BlockAddr &ad, auto G) {
        constexpr int g = decltype(G)::value;
        if constexpr (g < 8) {
            if constexpr (OFF32) g_n[g] = ld4(a.P + (ad.goff + 16u * g));
            else g_n[g] = ld4(ad.pj + 16 * g);
        } else {
            if (g == 8 && !DEC) mk_n = a.mask[ad.jj];
            e_n[g - 8] = ld4(ad.src + (eoff + 16u * (g - 8)));      // c = 2 step + half: columns 32 step + 16 half + 4 q
        }
    };
    auto issue_block = [&](int ii, int bb, int j) {
        const BlockAddr ad = block_addr(ii, bb, j);
        static_for<0, 16>([&](auto G) { issue_piece(ad, G); });
    };
    f4 g0_n = f4{0.f, 0.f, 0.f, 0.f};
    float mi_n = 0.f;
    auto issue_self = [&](int ii) {
        if (lane < 32) g0_n = ld4(a.P + (size_t)__builtin_amdgcn_readfirstlane(ii) * 256 + 4 * lane);
        mi_n = (a.mask + __builtin_amdgcn_readfirstlane(ii))[0];
    };

    int j_cur = idx_of(i, 0), j_nxt = idx_of(i, 1);
    issue_block(i, 0, j_cur);
    issue_self(i);
    // the NEXT block's B operand: its e rows are split into planes behind MFMA groups of the current block's GEMM 2 (round 6; until
    // then at the top of their own block, 80 vector instructions nothing else of the wavefront could hide)
    u4 xn[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) split_pair(e_n[2 * c], e_n[2 * c + 1], xn[c]);
    f4 sum[8];
    float mi = 0.f, cnt = 0.f;
    mark(-1);
    unsigned long long c_begin = 0, w_begin = 0;
    if (PROF) { c_begin = __builtin_readcyclecounter(); w_begin = wall_clock64(); }
    for (;;) {                                                  // residues of this wavefront
        if (lane < 32) st4(g0s + 4 * lane, g0_n);
        mi = mi_n;
        cnt = 0.f;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) sum[cb] = f4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // g0s is read back by this wavefront only (LDS is in order per wavefront)
        __builtin_amdgcn_wave_barrier();
        const bool more = t_res + 1 < nt;                       // this wavefront has another residue
        const int inx = more ? res_at(t_res + 1) : i;
#pragma unroll 1
        for (int b = 0; b < 3; ++b) {                           // its three 16-row blocks: the loop tools/isa_counts.py counts
            // ---- the block's operands (requested one block ago)
            u4 x[4][2];
            f4 acc[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) { x[c][0] = xn[c][0]; x[c][1] = xn[c][1]; }
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) acc[cb] = DEC ? g_n[cb] : g_n[cb] + ld4(g0s + 16 * cb + uq);
            // consumed HERE, in front of the requests that refill e_n / g_n: left free, hipcc sinks these into the block of their first
            // use (behind the requests), the old and the new operands are live together and every one of them is copied at the back edge
#pragma unroll
            for (int c = 0; c < 4; ++c) { asm volatile("" : "+v"(x[c][0]), "+v"(x[c][1])); }
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) touch(acc[cb]);
            mark(0);
            const float ma = j_cur < 0 ? 0.f : (DEC ? 1.f : mi * mk_n);
            // ---- requests: the next block's operands, the list entry of the block after it. UNCONDITIONAL (the last block of a
            // wavefront asks for its own operands again): a request under `if (more)` makes every operand register a phi that hipcc
            // copies around the loop — 64 v_mov_b64 per block in the first build of this kernel
            const bool last_b = b == 2;
            const bool has1 = !last_b || more;
            const int i1 = has1 ? (last_b ? inx : i) : i, b1 = has1 ? (last_b ? 0 : b + 1) : b;
            const BlockAddr ad1 = block_addr(i1, b1, has1 ? j_nxt : j_cur);
            // ---- the chain; the 16 request pieces ride behind the MFMA groups of GEMM 1
            mark(1);
            // (pieces 2 k and 2 k + 1 read the two halves of the same sixteen 128-byte lines; issuing them together behind every other
            //  group, so that the second finds the lines in the L1, was measured: nil — 0.1851 / 0.1857 against 0.1849 / 0.1844 ms)
            mma_wave_lds(wl1, x, acc, [&](auto G) { issue_piece(ad1, G); });
            if (last_b) issue_self(i1);
            {
                const bool last_b1 = b1 == 2;
                const int i2 = last_b1 ? inx : i1, b2 = last_b1 ? 0 : b1 + 1;          // (b1 == 2 only when the next block is of THIS residue)
                const bool has2 = has1 && (!last_b1 || more);
                j_cur = j_nxt;
                j_nxt = idx_of(has2 ? i2 : i1, has2 ? b2 : b1);
            }
            mark(2);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f4 v0 = acc[2 * c], v1 = acc[2 * c + 1];
                if (DEC) {
                    v0 = ld4(g0s + 32 * c + uq) + mi * v0;
                    v1 = ld4(g0s + 32 * c + 16 + uq) + mi * v1;
                }
                split_pair(gelu4(v0), gelu4(v1), x[c]);
            }
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) acc[cb] = ld4(s_b2 + 16 * cb + uq);
            mark(3);
            mma_wave_lds(wl2, x, acc, [&](auto G) {              // the next block's e rows (requested behind GEMM 1) -> planes
                constexpr int g = decltype(G)::value;
                if constexpr (g % 3 == 0 && g >= 3 && g <= 12) split_pair(e_n[2 * (g / 3 - 1)], e_n[2 * (g / 3 - 1) + 1], xn[g / 3 - 1]);
            });
            mark(4);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const f4 g = gelu4(acc[cb]);
                sum[cb] = f4{__builtin_fmaf(g.x, ma, sum[cb].x), __builtin_fmaf(g.y, ma, sum[cb].y), __builtin_fmaf(g.z, ma, sum[cb].z), __builtin_fmaf(g.w, ma, sum[cb].w)};
            }
            cnt += ma;
            mark(5);
        }
        // ---- sums over the 16 rows of a lane group (DPP row_shr scan: lane n = 15 ends up with the total)
#define TM_ROW_SCAN(x)                                                                                  \
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));  \
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));  \
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));  \
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
        // the totals (lane n = 15 of every lane group: 8 x 4 columns) meet in this wavefront's LDS row — the residue's own projection,
        // consumed by now — and leave as ONE 512-byte store
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            f4 t = sum[cb];
#pragma unroll
            for (int k = 0; k < 4; ++k) { float v = t[k]; TM_ROW_SCAN(v) t[k] = v; }
            if (n == 15) st4(g0s + 16 * cb + uq, t);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < 32) st4(a.Ssum + (size_t)__builtin_amdgcn_readfirstlane(i) * TM_H + 4 * lane, ld4(g0s + 4 * lane));
        TM_ROW_SCAN(cnt)
#undef TM_ROW_SCAN
        if (lane == 15) a.cnt[i] = cnt;
        mark(6);
        if (!more) break;
        ++t_res;
        i = inx;
    }
    if (PROF && tm_bid() == 0 && tm_tid() == (TM_PROF_TID & ~63)) {       // shader cycles and 100 MHz ticks of the loop: the clock under THIS load
#pragma unroll
        for (int k = 0; k < 8; ++k) prof[k] = t_acc[k];
        prof[8] = __builtin_readcyclecounter() - c_begin;
        prof[9] = wall_clock64() - w_begin;
    }
    if (PROF && (tm_bid() == 0 || tm_bid() == (int)gridDim.x - 1) && (tm_tid() & 63) == 0)                // every wavefront's loop, first and last workgroup
        prof[(tm_bid() == 0 ? 16 : 24) + (tm_tid() >> 6)] = __builtin_readcyclecounter() - c_begin;
}

int launch_msg_split(int mode, bool dec, const float *W1e, int ld1, const float *W2, const float *b2, const float *P,
                     const float *hE, const int32_t *E_idx, const float *mask, int64_t T, float *Ssum, float *cnt,
                     hipStream_t st) {
    const bool h2 = mode == TM_MM_F16X2;
    MsgArgsB a{W1e, ld1, W2, b2, P, hE, E_idx, mask, Ssum, cnt, (int)T, h2 ? tm_find_wimg(W1e) : nullptr, h2 ? tm_find_wimg(W2) : nullptr,
               h2 ? tm_find_wimgp(W1e) : nullptr, h2 ? tm_find_wimgp(W2) : nullptr, 0};
    const int64_t cap = tm_num_cus();
    const bool off32 = T < ((int64_t)1 << 22);       // projection table < 4 GB: 32-bit gather offsets
    // f16x2, large launches: whole multiples of 8 residues per workgroup go one wavefront per residue, the rest (< 8 per workgroup) to
    // the 8-wavefront form behind it — same K order, same summation order, same bits (tests: config 3 against single-protein forwards)
    static const int wave_min = TM_DBG_INT("TMPNN_MSG_WAVE_MIN", TM_MSG_WAVE_MIN);   // (debug library: 0 = from one residue per wavefront)
    int64_t Tw = 0;
    if (h2 && a.imgp1 && a.imgp2 && cap % 8 == 0) {
        const int64_t q = T / (8 * cap);
        if (q >= (wave_min > 0 ? wave_min : 1)) Tw = q * 8 * cap;
        else if (wave_min == 0 && T > 0) Tw = T;      // (debug switch: everything, whatever the size)
    }
    if (Tw > 0) {
        a.T = (int)Tw;
#ifdef TMPNN_DEBUG_BUILD
        static const bool wprof = TM_DBG_FLAG("TMPNN_MSG_PROF", false);
        if (wprof && dec) {                          // debug build: phase timing of one wavefront of workgroup 0 (synchronises!)
            static unsigned long long *d_prof = nullptr;
            if (!d_prof) (void)hipMalloc(&d_prof, 32 * sizeof(unsigned long long));
            (void)hipMemsetAsync(d_prof, 0, 32 * sizeof(unsigned long long), st);
            msg8_wave_kernel<true, false, true><<<(int)cap, 512, 0, st>>>(a, d_prof);
            unsigned long long h[32];
            (void)hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "dec_msg wave phases (cycles, one wavefront of wg 0, all its blocks): operands %llu requests %llu gemm1 %llu gelu+split %llu gemm2 %llu gelu+mask %llu ksum+store %llu; loop %llu cycles in %llu ticks of 100 MHz = %.3f GHz\n",
                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[8], h[9], h[9] ? h[8] / (h[9] * 10.0) : 0.0);
            fprintf(stderr, "dec_msg wave loops (cycles, the eight wavefronts of wg 0): %llu %llu %llu %llu | %llu %llu %llu %llu; of the last wg: %llu %llu %llu %llu | %llu %llu %llu %llu\n",
                    h[16], h[17], h[18], h[19], h[20], h[21], h[22], h[23], h[24], h[25], h[26], h[27], h[28], h[29], h[30], h[31]);
        } else
#endif
        if (off32) {
            if (dec) msg8_wave_kernel<true, true><<<(int)cap, 512, 0, st>>>(a);
            else msg8_wave_kernel<false, true><<<(int)cap, 512, 0, st>>>(a);
        } else if (dec) msg8_wave_kernel<true, false><<<(int)cap, 512, 0, st>>>(a);
        else msg8_wave_kernel<false, false><<<(int)cap, 512, 0, st>>>(a);
        if (Tw == T) return tm_check_launch(dec ? "dec_msg_wave" : "enc_msg_wave");
        a.i0 = (int)Tw;
        a.T = (int)(T - Tw);
    }
    const int grid = (int)(a.T < cap ? a.T : cap);
    if (mode == TM_MM_BF16X3) {
        if (off32) {
            if (dec) msg8_rp_kernel<SplitBF3, true, false, true><<<grid, 512, 0, st>>>(a);
            else msg8_rp_kernel<SplitBF3, false, false, true><<<grid, 512, 0, st>>>(a);
        } else if (dec) msg8_rp_kernel<SplitBF3, true><<<grid, 512, 0, st>>>(a);
        else msg8_rp_kernel<SplitBF3, false><<<grid, 512, 0, st>>>(a);
    } else {
#ifdef TMPNN_DEBUG_BUILD
        static const bool prof = TM_DBG_FLAG("TMPNN_MSG_PROF", false);
#else
        constexpr bool prof = false;
#endif
        if (prof && dec && Tw == 0) {                // debug build: phase timing of workgroup 0 (synchronises!)
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
        } else if (off32) {
            if (dec) msg8_rp_kernel<SplitH2, true, false, true><<<grid, 512, 0, st>>>(a);
            else msg8_rp_kernel<SplitH2, false, false, true><<<grid, 512, 0, st>>>(a);
        } else if (dec) msg8_rp_kernel<SplitH2, true><<<grid, 512, 0, st>>>(a);
        else msg8_rp_kernel<SplitH2, false><<<grid, 512, 0, st>>>(a);
    }
    return tm_check_launch(dec ? "dec_msg_split" : "enc_msg_split");
}
