// Edge update of an encoder layer (protein_mpnn_utils.py:826-839), f16x2, large launches: one WAVEFRONT per 16-row block of a residue's
// edge tile, one wavefront per SIMD (round 6). MEASURED AND NOT SHIPPED — the kernel exists in the debug library only
// (-DTMPNN_DEBUG_BUILD, TMPNN_EDGE_WAVE_MIN=6): within +-3 % of the 8-wavefront form in one call (0.283-0.299 against 0.293-0.307 ms per
// launch on the bench batch, docs/NOTEBOOK.md 10.2, profiles/r06_ab_edge.txt), and its K order and LayerNorm summation order differ from
// the shipped forms' in the last bits (2e-6 on the hidden states), so shipping it would have meant re-deriving those for nothing.
// What it established: a lone wavefront issues a v_mfma_f32_16x16x32_f16 every 16.7 cycles whatever the distance between two uses of
// an accumulator and hides TWO independent VALU instructions behind it, every further one costs ~4.5 cycles (tools/probe/dep_probe,
// shadow_probe); the unit's 288 MFMAs + ~1 130 vector instructions + 169 LDS reads therefore cost >= 9 k cycles where the 8-wavefront
// form's tile (three units on four SIMDs) takes 10.1 k.
//
// The 8-wavefront form (enc_edge8_rp_kernel, tmpnn_edge.hip) splits the 128 output columns of each of the three GEMMs over the wavefronts
// of a workgroup: every GEMM starts behind a barrier, the activations make an LDS round trip per GEMM, and the two wavefronts of a SIMD
// run matrix and vector phases in lock step (docs/NOTEBOOK.md 9.2-9.5: pipe times add). Here a wavefront owns 16 edge rows through
//     e (global) -> B operand -> GEMM 1 -> GELU -> B operand -> GEMM 2 -> GELU -> B operand -> GEMM 3 -> residual -> LayerNorm -> store
// and the activations never leave its registers: v_mfma_f32_16x16x32_f16 with A = weights, B = activations leaves lane (n, q) holding
// output columns 16 cb + 4 q + {0..3} of edge row n, and with K-permuted weight images (perm_c4, tmpnn_split.h) the GELU-ed, split
// accumulator blocks 2 c and 2 c + 1 ARE the B operand of step c of the next GEMM (as in msg8_wave_kernel). A row's 128 outputs sit in
// 4 lanes x 32 registers: the LayerNorm statistics are 32 local terms and two permlane swaps, no LDS, no barrier.
// Weights: three 64 KB fragment images do not fit the 160 KB LDS — W11e and W12 sit in LDS (conflict-free 1 KB fragment reads), W13 in
// REGISTERS: 256 per lane, which a workgroup of FOUR wavefronts (one per SIMD, 512 registers each: VGPRs + AGPRs, MFMA operands may be
// either) can afford. With one wavefront per SIMD nothing else hides a wavefront's vector work, so the block is software-pipelined by
// hand: GEMM 1 runs column-block-major, GEMM 2 / 3 step-major, and the GELU + split of accumulator pair c is issued value by value
// between the MFMA triples of the stages that do not need it yet (MI355X_MICROARCH.md: a lone wavefront hides a few single-issue
// instructions per MFMA; never between two MFMAs of ONE accumulator). No workgroup barrier after the prologue.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "tmpnn_split.h"
#include "tmpnn_internal.h"

// gelu2 (tmpnn_common.h, TM_GELU_NAN3 form) one value at a time with scalar fmas: the same operations in the same order — the same bits.
// Beside MFMAs a packed fp32 op costs a lone wavefront more than the two scalar ones it replaces (MI355X_MICROARCH.md, filler prices).
__device__ __forceinline__ float gelu1n(float x) {
#if TM_ABL_NOGELU
    return x;
#endif
    const float t = __builtin_elementwise_minimum(fabsf(x), 5.656854249f);
    float p = __builtin_fmaf(3.309543916e-05f, t, -7.692427171e-04f);
    p = __builtin_fmaf(p, t, 8.080792133e-03f);
    p = __builtin_fmaf(p, t, -5.341222090e-02f);
    p = __builtin_fmaf(p, t, -4.587708865e-01f);
    p = __builtin_fmaf(p, t, -1.151201730e+00f);
    p = __builtin_fmaf(p, t, -9.999930581e-01f);
    return __builtin_fmaf(-t, __builtin_amdgcn_exp2f(p), __builtin_elementwise_maximum(x, 0.f));
}

#ifdef TMPNN_DEBUG_BUILD
#define TM_HF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0)
// ... with the weight fragment in AGPRs (inline asm: hipcc's MFMA selection would copy the fragment to VGPRs first). What the hazard
// recogniser cannot see inside the asm is handled here: FIRST (the first MFMA of a slab) = s_nop 1, the two wait states between a VALU
// write of x and an MFMA reading it (the riders between the MFMAs of a slab write neither x[step] nor an accumulator of GEMM 3);
// whoever reads `acc` with a non-MFMA instruction next must have issued mma_asm_settle() first.
template <bool FIRST>
__device__ __forceinline__ void mma_one_a(const u4 &w, const u4 &x, f4 &acc) {
#if TM_ABL_NOMFMA
    return;
#endif
    if constexpr (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
}
__device__ __forceinline__ void mma_asm_settle() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }

// The 16 slabs of the LDS-fed GEMMs (see the kernel): slab -> byte offset of its first fragment inside its image (slabs 0..7: W11e, 8..15: W12), first accumulator, step
struct EwSlab { int off, cb0, step; };
__device__ __forceinline__ constexpr EwSlab ew_slab(int sl) {
    return sl < 8 ? EwSlab{4 * (sl >> 2) * 8192 + (sl & 3) * 2048, 4 * (sl >> 2), sl & 3}
                  : EwSlab{4 * ((sl - 8) & 1) * 8192 + ((sl - 8) >> 1) * 2048, 4 * ((sl - 8) & 1), (sl - 8) >> 1};
}

template <bool OFF32, bool PROF = false>
__global__ __launch_bounds__(256, 1) void enc_edge_wave_kernel(EdgeArgsB a, unsigned long long *prof = nullptr) {
    __shared__ __attribute__((aligned(16))) char sW[2 * TM_WIMG_BYTES];
    __shared__ __attribute__((aligned(16))) float s_pi[4][TM_H];      // the residue's own projection row, per wavefront
    __shared__ __attribute__((aligned(16))) float s_par[4][TM_H];     // b12, b13, norm3 weight, norm3 bias
    const int tid = tm_tid(), lane = tid & 63, wv = tm_wave(tid), n = lane & 15, q = lane >> 4;
    for (int o = tid * 16; o < TM_WIMG_BYTES; o += 256 * 16) {
        *reinterpret_cast<u4 *>(sW + o) = *reinterpret_cast<const u4 *>(a.imgp11 + o);
        *reinterpret_cast<u4 *>(sW + TM_WIMG_BYTES + o) = *reinterpret_cast<const u4 *>(a.imgp12 + o);
    }
    if (tid < 128) {
        const float *src = tid < 32 ? a.b12 : tid < 64 ? a.b13 : tid < 96 ? a.g3 : a.be3;
        st4(&s_par[tid >> 5][4 * (tid & 31)], ld4(src + 4 * (tid & 31)));
    }
    // W13, unit s = 8 step + cb of GEMM 3, both planes: 256 registers per lane, pinned to the ACCUMULATION half of the register file
    // (inline asm, "a" constraints: left to itself hipcc parks them there as spills and copies each fragment back with four
    // v_accvgpr_read in front of its use; an MFMA reads an A operand from an AGPR directly)
    u4 w13[32][2];
    {
        const char *src = a.imgp13 + lane * 16;
#pragma unroll
        for (int s = 0; s < 32; ++s)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(w13[s][p]) : "v"(src + (s & 7) * 8192 + (s >> 3) * 2048 + p * 1024) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                                   // the only barrier of the kernel
    const char *wl = sW + 16 * lane;
    unsigned img2 = TM_WIMG_BYTES;                                     // second image: its own base register (a ds_read offset has 16 bits;
    asm volatile("" : "+v"(img2));                                     //  folded into ONE base, every read of it costs a v_or for the address)
    const char *wl2 = wl + img2;
    float *pis = s_pi[wv];
    const unsigned uq = 4u * (unsigned)q;
    const unsigned eoff = (unsigned)(n * TM_H) + uq;                   // this lane's offset inside a 16-row block of e

    // units of this workgroup: (residue k of its range, block b) <-> v = 3 k + b; wavefront w takes v = w, w + 4, ...
    const TileRange tr = xcd_tile_range(a.T);
    const int nres = tr.begin < tr.end ? (tr.end - tr.begin + tr.step - 1) / tr.step : 0;
    const int nunits = 3 * nres;
    int v = wv;
    if (v >= nunits) return;
    auto res_of = [&](int vv) { return __builtin_amdgcn_readfirstlane(tr.begin + (vv / 3) * tr.step); };
    auto blk_of = [&](int vv) { return __builtin_amdgcn_readfirstlane(vv % 3); };
    auto idx_of = [&](int vv) { return (a.E_idx + ((size_t)res_of(vv) * TM_KS + 16 * blk_of(vv)))[(unsigned)n]; };

    f4 e_n[8], g_n[8];
    f4 pi_n = f4{0.f, 0.f, 0.f, 0.f};
    // operands of a unit as 17 pieces (one global request each): 0..7 the e rows (block c: columns 16 c + 4 q, c = 2 step + half) — first,
    // loads return in order and the e rows are wanted first (split behind the end of GEMM 3) —, 8..15 the gathered projection row
    // P[j, 128 + 16 cb + 4 q], 16 the residue's own projection row. Addresses are formed once per unit.
    struct UnitAddr { unsigned goff; const float *pj; const float *src; const float *pi; };
    auto unit_addr = [&](int vv, int j) {
        const int ii = res_of(vv), bb = blk_of(vv);
        const int jj = j < 0 ? ii : j;
        UnitAddr r;
        r.goff = (unsigned)jj * 256u + (128u + uq);
        r.pj = a.P + (size_t)jj * 256 + 128 + uq;
        r.src = a.hE + ((size_t)ii * TM_KS + 16 * bb) * TM_H;
        r.pi = a.P + (size_t)ii * 256 + 4 * lane;
        return r;
    };
    auto issue_piece = [&](const UnitAddr &ad, auto G) {
        constexpr int g = decltype(G)::value;
        if constexpr (g < 8) {
            e_n[g] = ld4(ad.src + (eoff + 16u * g));
        } else if constexpr (g < 16) {
            if constexpr (OFF32) g_n[g - 8] = ld4(a.P + (ad.goff + 16u * (g - 8)));
            else g_n[g - 8] = ld4(ad.pj + 16 * (g - 8));
        } else {
            if (lane < 32) pi_n = ld4(ad.pi);
        }
    };
    auto vclamp = [&](int vv) { return vv < nunits ? vv : v; };       // beyond the last unit: ask for the current one again (unconditional requests)

    int j_cur = idx_of(v), j_nxt = idx_of(vclamp(v + 4));
    {
        const UnitAddr ad0 = unit_addr(v, j_cur);
        static_for<0, 17>([&](auto G) { issue_piece(ad0, G); });
    }
    // the NEXT unit's GEMM-1 operand: its e rows are split into planes behind the last MFMAs of the current unit's GEMM 3
    u4 xn[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) split_pair(e_n[2 * c], e_n[2 * c + 1], xn[c]);
    // TMPNN_EDGE_PROF=1 (debug library): phase timing of wavefront 0 of workgroup 0. Scalar registers only (s_memtime + SALU adds; written
    // out once behind the loop): a timer that does a global read-modify-write per mark waits for every request in flight at every mark.
    unsigned t_last = 0, t_acc[40] = {};
    auto mark = [&](int k) {
        if constexpr (PROF) {
            __builtin_amdgcn_sched_barrier(0);              // (s_memtime is no scheduling barrier by itself: the phases would smear)
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            if (k >= 0) t_acc[k] += t - t_last;
            t_last = t;
        }
    };
    unsigned long long c_begin = 0, w_begin = 0;
    if (PROF) { c_begin = __builtin_readcyclecounter(); w_begin = wall_clock64(); }
    mark(-1);
#pragma unroll 1
    for (;;) {
        const int ii = res_of(v), bb = blk_of(v);
        // ---- this unit's operands (requested one unit ago)
        if (lane < 32) st4(pis + 4 * lane, pi_n);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // pis is read back by this wavefront only
        __builtin_amdgcn_wave_barrier();
        u4 x[4][2], x2[4][2];
        f4 acc[8], acc2[8], e_cur[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) { x[c][0] = xn[c][0]; x[c][1] = xn[c][1]; }
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            e_cur[cb] = e_n[cb];
            acc[cb] = ld4(pis + 16 * cb + uq) + g_n[cb];
        }
        const bool valid = j_cur >= 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) { asm volatile("" : "+v"(x[c][0]), "+v"(x[c][1])); }
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) { touch(acc[cb]); touch(e_cur[cb]); }
        mark(0);
        // ---- requests: the next unit's operands ride behind the MFMAs of GEMM 1's first half, one global request per slot (17 in a row
        // cost the wavefront ~55-100 cycles of issue each), the list entry of the unit after it behind the last of them
        const int v1 = vclamp(v + 4), v2 = vclamp(v + 8);
        const UnitAddr ad1 = unit_addr(v1, v1 != v ? j_nxt : j_cur);
        j_cur = v1 != v ? j_nxt : j_cur;
        mark(1);

        // ---- the chain. A SLAB = the twelve MFMAs of four accumulators x one 32-deep step, issued pass by pass (l h of the four, h l of
        // the four, h h of the four): no MFMA follows another on the same accumulator closer than four issues — a lone wavefront pays
        // ~33 cycles for a dependent issue against 17 for an independent one (the first build of this kernel, column blocks one after the
        // other: 41-49 cycles per MFMA). Per accumulator the order is still step by step, l h, h l, h h: the bits of SplitH2::mma.
        // LDS slabs 0..7 = GEMM 1 (accumulators 0..3 through all four steps, then 4..7: the first GELU pairs are ready at half time),
        // 8..15 = GEMM 2 step-major (its operand arrives pair by pair); fragments double-buffered slab by slab.
        // Fragment registers are recycled plane by plane: the l plane of a slab is dead behind its first pass, the h plane behind its
        // third — the l plane of slab sl + 2 is requested behind pass 0 of slab sl, its h plane behind pass 2 (16-20 MFMAs ahead of
        // their first use with the 64 registers of two slabs; requested slab by slab, 12 ahead, a bare slab took 300 cycles against 200).
        u4 fr[2][4][2];
        auto request_plane = [&](auto SL, auto P) {
            constexpr int sl = decltype(SL)::value, pl = decltype(P)::value;
            if constexpr (sl < 16) {
                constexpr EwSlab u = ew_slab(sl);
#if defined(TM_ABL_WAVE_NOLDS)
                if (sl > 1) return;
#endif
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    fr[sl & 1][k][pl] = *reinterpret_cast<const u4 *>((sl < 8 ? wl : wl2) + u.off + k * 8192 + pl * 1024);
            }
        };
        request_plane(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        request_plane(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        request_plane(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        request_plane(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        // GELU + split of value k (0..7) of accumulator pair c of `src` into the B operand dst (complete behind k = 7)
        float gt[8];
        auto gelu_val = [&](const f4 (&src)[8], u4 (&dst)[2], auto C, auto K) {
            constexpr int k = decltype(K)::value, c = decltype(C)::value;
            gt[k] = gelu1n(src[2 * c + (k >> 2)][k & 3]);
            if constexpr (k == 7) {
                unsigned p0[2], p1[2], p2[2], p3[2];
                SplitH2::split2(f2{gt[0], gt[1]}, p0);
                SplitH2::split2(f2{gt[2], gt[3]}, p1);
                SplitH2::split2(f2{gt[4], gt[5]}, p2);
                SplitH2::split2(f2{gt[6], gt[7]}, p3);
                dst[0] = u4{p0[0], p1[0], p2[0], p3[0]};
                dst[1] = u4{p0[1], p1[1], p2[1], p3[1]};
            }
        };
        // ride(R), R = 0..3: the vector work that goes behind MFMAs 3 R + 2 of a slab (between MFMAs of DIFFERENT accumulators)
        auto slab_lds = [&](auto SL, const u4 (&xin)[4][2], f4 (&ac)[8], auto &&ride) {
            constexpr int sl = decltype(SL)::value;
            constexpr EwSlab u = ew_slab(sl);
            static_for<0, 12>([&](auto M) {
                constexpr int m = decltype(M)::value, pass = m >> 2, k = m & 3;
                __builtin_amdgcn_sched_barrier(0);
#if !TM_ABL_NOMFMA
                ac[u.cb0 + k] = TM_HF(fr[sl & 1][k][pass == 0 ? 1 : 0], xin[u.step][pass == 1 ? 1 : 0], ac[u.cb0 + k]);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m == 3) request_plane(std::integral_constant<int, sl + 2>{}, std::integral_constant<int, 1>{});
                if constexpr (m == 11) request_plane(std::integral_constant<int, sl + 2>{}, std::integral_constant<int, 0>{});
                if constexpr (m % 3 == 2) ride(std::integral_constant<int, m / 3>{});
            });
            __builtin_amdgcn_sched_barrier(0);
            mark(8 + sl);
        };
        auto slab_reg = [&](auto S, auto Q, auto &&ride) {                 // GEMM 3: step S, accumulators 4 Q .. 4 Q + 3, W13 from the AGPRs
            constexpr int step = decltype(S)::value, cb0 = 4 * decltype(Q)::value;
            static_for<0, 12>([&](auto M) {
                constexpr int m = decltype(M)::value, pass = m >> 2, k = m & 3;
                __builtin_amdgcn_sched_barrier(0);
                mma_one_a<m == 0>(w13[8 * step + cb0 + k][pass == 0 ? 1 : 0], x[step][pass == 1 ? 1 : 0], acc[cb0 + k]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m % 3 == 2) ride(std::integral_constant<int, m / 3>{});
            });
            __builtin_amdgcn_sched_barrier(0);
            mark(24 + 2 * step + cb0 / 4);
        };
        auto nothing = [&](auto) {};
#define TM_IC(n) std::integral_constant<int, n>{}
        // riders: value 4 (slab & 1) + R of GELU pair C of `src` into dst
#define TM_RIDE(src, dst, C, SLAB) [&](auto R) { gelu_val(src, dst, TM_IC(C), std::integral_constant<int, 4 * ((SLAB) & 1) + decltype(R)::value>{}); }
        // GEMM 1; GELU 1 of pairs 0, 1 (accumulators 0..3, final after slab 3) rides behind slabs 4..7
#define TM_REQ(SLAB) [&](auto R) { issue_piece(ad1, std::integral_constant<int, 4 * (SLAB) + decltype(R)::value>{}); }
        slab_lds(TM_IC(0), x, acc, TM_REQ(0));
        slab_lds(TM_IC(1), x, acc, TM_REQ(1));
        slab_lds(TM_IC(2), x, acc, TM_REQ(2));
        slab_lds(TM_IC(3), x, acc, [&](auto R) {
            issue_piece(ad1, std::integral_constant<int, 12 + decltype(R)::value>{});
            if constexpr (decltype(R)::value == 3) {
                issue_piece(ad1, TM_IC(16));
                j_nxt = idx_of(v2);
            }
        });
#undef TM_REQ
        slab_lds(TM_IC(4), x, acc, TM_RIDE(acc, x2[0], 0, 4));
        slab_lds(TM_IC(5), x, acc, TM_RIDE(acc, x2[0], 0, 5));
        slab_lds(TM_IC(6), x, acc, TM_RIDE(acc, x2[1], 1, 6));
        slab_lds(TM_IC(7), x, acc, TM_RIDE(acc, x2[1], 1, 7));
        // GEMM 2 (steps 0, 1 need pairs 0, 1 only); GELU 1 of pairs 2, 3 rides behind steps 0 and 1
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc2[cb] = ld4(s_par[0] + 16 * cb + uq);
        mark(2);
        slab_lds(TM_IC(8), x2, acc2, TM_RIDE(acc, x2[2], 2, 8));
        slab_lds(TM_IC(9), x2, acc2, TM_RIDE(acc, x2[2], 2, 9));
        slab_lds(TM_IC(10), x2, acc2, TM_RIDE(acc, x2[3], 3, 10));
        slab_lds(TM_IC(11), x2, acc2, TM_RIDE(acc, x2[3], 3, 11));
        slab_lds(TM_IC(12), x2, acc2, nothing);
        slab_lds(TM_IC(13), x2, acc2, nothing);
        slab_lds(TM_IC(14), x2, acc2, nothing);
        // accumulators 0..3 of GEMM 2 are final behind slab 14: GELU 2 of pair 0 rides behind slab 15, two values per slot
        slab_lds(TM_IC(15), x2, acc2, [&](auto R) {
            gelu_val(acc2, x[0], TM_IC(0), std::integral_constant<int, 2 * decltype(R)::value>{});
            gelu_val(acc2, x[0], TM_IC(0), std::integral_constant<int, 2 * decltype(R)::value + 1>{});
        });
        // GEMM 3 from the register-resident W13, step-major; GELU 2 of pair c + 1 rides behind step c
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc[cb] = ld4(s_par[1] + 16 * cb + uq);
        mark(3);
        slab_reg(TM_IC(0), TM_IC(0), TM_RIDE(acc2, x[1], 1, 0));
        slab_reg(TM_IC(0), TM_IC(1), TM_RIDE(acc2, x[1], 1, 1));
        slab_reg(TM_IC(1), TM_IC(0), TM_RIDE(acc2, x[2], 2, 0));
        slab_reg(TM_IC(1), TM_IC(1), TM_RIDE(acc2, x[2], 2, 1));
        slab_reg(TM_IC(2), TM_IC(0), TM_RIDE(acc2, x[3], 3, 0));
        slab_reg(TM_IC(2), TM_IC(1), TM_RIDE(acc2, x[3], 3, 1));
        // ... and the next unit's e rows -> planes behind the last step (x2 is free: its registers take them)
        slab_reg(TM_IC(3), TM_IC(0), [&](auto R) { if constexpr (decltype(R)::value >= 2) split_pair(e_n[2 * (decltype(R)::value - 2)], e_n[2 * (decltype(R)::value - 2) + 1], xn[decltype(R)::value - 2]); });
        slab_reg(TM_IC(3), TM_IC(1), [&](auto R) { if constexpr (decltype(R)::value < 2) split_pair(e_n[4 + 2 * decltype(R)::value], e_n[5 + 2 * decltype(R)::value], xn[2 + decltype(R)::value]); });
#undef TM_RIDE
#undef TM_IC
        mma_asm_settle();
        mark(4);

        // ---- residual, LayerNorm over the row (4 lanes x 32 values), store in place
        float s4[8];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            acc[cb] = e_cur[cb] + acc[cb];
            s4[cb] = (acc[cb].x + acc[cb].y) + (acc[cb].z + acc[cb].w);
        }
        float mean = 0.f, rstd = 1.f;
#if !TM_ABL_NOLN
        mean = swap_add32(swap_add16(((s4[0] + s4[1]) + (s4[2] + s4[3])) + ((s4[4] + s4[5]) + (s4[6] + s4[7])))) * (1.0f / 128.0f);
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const f4 d = acc[cb] - mean;
            s4[cb] = __builtin_fmaf(d.w, d.w, __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x)));
        }
        const float m2 = swap_add32(swap_add16(((s4[0] + s4[1]) + (s4[2] + s4[3])) + ((s4[4] + s4[5]) + (s4[6] + s4[7]))));
        rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(m2, 1.0f / 128.0f, 1e-5f));
#endif
        float *dst = a.hE + ((size_t)ii * TM_KS + 16 * bb) * TM_H;
        touch(e_n[0]);                                                 // the next unit's operands have long arrived: take their vmcnt wait in front of the stores
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const f4 g4 = ld4(s_par[2] + 16 * cb + uq), be4 = ld4(s_par[3] + 16 * cb + uq);
            const f4 s = g4 * rstd;
            const f4 t = f4{__builtin_fmaf(-mean, s.x, be4.x), __builtin_fmaf(-mean, s.y, be4.y), __builtin_fmaf(-mean, s.z, be4.z), __builtin_fmaf(-mean, s.w, be4.w)};
            const f4 y = f4{__builtin_fmaf(acc[cb].x, s.x, t.x), __builtin_fmaf(acc[cb].y, s.y, t.y), __builtin_fmaf(acc[cb].z, s.z, t.z), __builtin_fmaf(acc[cb].w, s.w, t.w)};
            // rows without a neighbour keep the zeros the featurizer wrote: store zeros again (no divergent branch)
            st4(dst + (eoff + 16u * cb), valid ? y : f4{0.f, 0.f, 0.f, 0.f});
        }
        mark(5);
        if (v + 4 >= nunits) break;
        v += 4;
    }
    if (PROF && tm_bid() == 0 && tid == 0) {
#pragma unroll
        for (int k = 0; k < 40; ++k) prof[k] = t_acc[k];
        prof[40] = __builtin_readcyclecounter() - c_begin;
        prof[41] = wall_clock64() - w_begin;
    }
}

// Work unit = one 16-row block of one wavefront (a few microseconds): a launch of at least TMPNN_EDGE_WAVE_MIN blocks per wavefront goes
// here whole (at most one unit of imbalance). Debug library only; default: never.
bool enc_edge_wave_wanted(int64_t T) {
    static const int wave_min = TM_DBG_INT("TMPNN_EDGE_WAVE_MIN", -1);
    return wave_min >= 0 && tm_num_cus() % 8 == 0 && 3 * T >= (int64_t)(wave_min > 0 ? wave_min : 1) * 4 * tm_num_cus();
}
int launch_enc_edge_wave(const EdgeArgsB &a, int64_t T, hipStream_t st) {
    const int cap = tm_num_cus();
    static const bool prof = TM_DBG_FLAG("TMPNN_EDGE_PROF", false);
    if (prof) {                                      // debug build: phase timing of wavefront 0 of workgroup 0 (synchronises!)
        static unsigned long long *d_prof = nullptr;
        if (!d_prof) (void)hipMalloc(&d_prof, 48 * sizeof(unsigned long long));
        (void)hipMemsetAsync(d_prof, 0, 48 * sizeof(unsigned long long), st);
        enc_edge_wave_kernel<false, true><<<cap, 256, 0, st>>>(a, d_prof);
        unsigned long long h[48];
        (void)hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost);
        const long long res0 = T >= 8 * cap ? (T / 8 + cap / 8 - 1) / (cap / 8) : (T + cap - 1) / cap;           // residues of workgroup 0 (xcd_tile_range)
        fprintf(stderr, "enc_edge wave phases (cycles, wavefront 0 of wg 0, all its units): operands %llu addresses %llu acc2-init %llu acc3-init %llu settle %llu ln+store %llu; loop %llu cycles in %llu ticks of 100 MHz = %.3f GHz, %lld residues = %.0f cycles per tile (4 wavefronts, 3 units per tile)\n",
                h[0], h[1], h[2], h[3], h[4], h[5], h[40], h[41], h[41] ? h[40] / (h[41] * 10.0) : 0.0, res0, res0 ? (double)h[40] / res0 : 0.0);
        fprintf(stderr, "enc_edge wave slabs (cycles per unit; 12 MFMAs each): GEMM 1");
        const double nu = 3.0 * res0 / 4.0;
        for (int k = 8; k < 32; ++k) fprintf(stderr, "%s %.0f", k == 16 ? " | GEMM 2" : k == 24 ? " | GEMM 3" : "", h[k] / nu);
        fprintf(stderr, "\n");
        return tm_check_launch("enc_edge_wave");
    }
    if (T < ((int64_t)1 << 22)) enc_edge_wave_kernel<true><<<cap, 256, 0, st>>>(a);
    else enc_edge_wave_kernel<false><<<cap, 256, 0, st>>>(a);
    return tm_check_launch("enc_edge_wave");
}
#else
// the shipped library holds one form of the edge update per precision and launch-size class (tmpnn_edge.hip, tmpnn_edge_msg.hip)
bool enc_edge_wave_wanted(int64_t) { return false; }
int launch_enc_edge_wave(const EdgeArgsB &, int64_t, hipStream_t) { return tm_set_error(TMPNN_E_INVALID, "enc_edge_wave: debug library only"); }
#endif
