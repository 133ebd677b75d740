// "Wave-owns-tile" forms of the per-edge kernels (f16x2 arithmetic): message pass (encoder + decoder) and the encoder
// edge update. Reference semantics: EncLayer.forward (/root/reference/protein_mpnn_utils.py:816-839), DecLayer.forward
// (:859-880), decoder wiring (:1268-1273).
//
// The round-1 kernels split the 128 output columns of a 48 x 128 residue tile over the 8 wavefronts of a workgroup
// (weights in VGPRs, activations through LDS, 2-5 workgroup barriers per tile): all 8 wavefronts sit in the same phase,
// so the matrix pipe idles during every GELU/split phase and the VALU idles during every GEMM (MFMA 26 %, VALU ~50 %
// busy, a third of the wavefront cycles parked at barriers / s_waitcnt).
// Here the roles are swapped:
//   * the WEIGHTS live in LDS, pre-split into f16 planes and pre-arranged as ready-made MFMA A fragments
//     (1 KB per fragment plane: lane l reads its 16 bytes at l * 16 — linear, conflict-free ds_read_b128);
//   * every WAVEFRONT owns whole residue tiles and carries its activations in REGISTERS through all GEMMs of the
//     tile: the D layout of v_mfma_f32_16x16x32_f16 with A = weights (lane (m, q): 4 consecutive output features of
//     row m) is, after a fixed permutation of the output features that is folded into the LDS weight image, exactly
//     the B layout (lane (m, q): 8 consecutive input features of row m) of the next GEMM — no LDS round trip, no
//     cross-lane traffic, NO workgroup barrier in the main loop;
//   * the 8 wavefronts of a CU drift apart, so one wavefront's GELU/split (VALU) runs under its SIMD partner's MFMAs.
// A tile is processed as two passes (rows 0-31, then rows 32-47) to fit the 256-VGPR budget of 2 wavefronts per SIMD;
// an A fragment read feeds 6 / 3 MFMAs.
//
// Arithmetic: x = h + l with h = fp16(x), l = fp16(x - h) UNSCALED (l may be an fp16 subnormal; the matrix core does
// not flush them): |x - (h + l)| <= 2^-25 for |x| < 2, 2^-23 relative above. Product = h h + h l + l h in ONE fp32
// accumulator (three v_mfma_f32_16x16x32_f16 per 32-deep step). Needs |x| < 65504 like the scaled f16x2 form.
#include <stdlib.h>

#include "tmpnn_split.h"
#include "tmpnn_internal.h"

namespace {

constexpr int WT_FRAG_BYTES = 1024;                  // one A-fragment plane: 64 lanes x 16 B
constexpr int WT_PLANE_BYTES = 32 * WT_FRAG_BYTES;   // 8 column blocks x 4 k-steps: one 128 x 128 weight, one plane

// First of the 4 consecutive features lane quarter q holds in output block cb (D layout), chosen so that blocks
// (2c, 2c+1) of a lane are the 8 consecutive input features [32c + 8q, 32c + 8q + 8) of k-step c of the next GEMM.
__device__ __forceinline__ int wt_col(int cb, int q) { return 32 * (cb >> 1) + 8 * q + 4 * (cb & 1); }
// Output feature computed by A-operand row n (0..15) of block cb — the inverse view of wt_col.
__device__ __forceinline__ int wt_row(int cb, int n) { return 32 * (cb >> 1) + 8 * (n >> 2) + 4 * (cb & 1) + (n & 3); }

__device__ __forceinline__ void wt_split2(f2 x, unsigned &h, unsigned &l) {
    const h2 hh = __builtin_convertvector(x, h2);                            // v_cvt_pk_f16_f32 (RNE)
    const f2 r = x - __builtin_convertvector(hh, f2);                        // exact
    const h2 ll = __builtin_convertvector(r, h2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}

struct XFrag { u4 h, l; };     // B operand of one 32-deep k-step of one 16-row block: 8 features x 2 planes per lane

// features [8q', 8q'+8) of a row given as two f4 -> one B fragment
__device__ __forceinline__ XFrag wt_split8(f4 v0, f4 v1) {
    unsigned h[4], l[4];
    wt_split2(f2{v0.x, v0.y}, h[0], l[0]);
    wt_split2(f2{v0.z, v0.w}, h[1], l[1]);
    wt_split2(f2{v1.x, v1.y}, h[2], l[2]);
    wt_split2(f2{v1.z, v1.w}, h[3], l[3]);
    return XFrag{u4{h[0], h[1], h[2], h[3]}, u4{l[0], l[1], l[2], l[3]}};
}
// the 4 outputs of block cb go into half (cb & 1) of the next GEMM's k-step (cb >> 1) fragment
template <int HALF>
__device__ __forceinline__ void wt_put4(XFrag &x, f4 v) {
    unsigned h0, l0, h1, l1;
    wt_split2(f2{v.x, v.y}, h0, l0);
    wt_split2(f2{v.z, v.w}, h1, l1);
    // pins the GELU + split HERE in the side-effect chain: pure arithmetic is otherwise free to sink past the WT_FENCE()s
    // (instruction selection linearises it late), which piles up every block's raw accumulators and serialises the phases
    asm volatile("" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));
    if (HALF == 0) { x.h.x = h0; x.h.y = h1; x.l.x = l0; x.l.y = l1; }
    else { x.h.z = h0; x.h.w = h1; x.l.z = l0; x.l.w = l1; }
}

// keeps hipcc from moving code across column-block iterations (it otherwise hoists the loads / LDS reads of many
// iterations to the top and spills hundreds of VGPRs)
#define WT_FENCE() __builtin_amdgcn_sched_barrier(0)
// Loop-invariant small operands (biases, LayerNorm parameters) are re-read from L1 every tile: laundering the pointer
// once per tile keeps LICM from hoisting 100+ VGPRs worth of them out of the tile loop.
// (The result is typed as a GLOBAL pointer: a laundered generic pointer would turn every access into flat_load, which
// ticks lgkmcnt as well and makes hipcc wait with vmcnt(0) — draining every prefetch in flight.)
typedef const __attribute__((address_space(1))) float *gfloat_p;
typedef const __attribute__((address_space(1))) char *gchar_p;
__device__ __forceinline__ gfloat_p wt_launder(const float *p) {
    asm volatile("" : "+s"(p));
    return (gfloat_p)p;
}
__device__ __forceinline__ gchar_p wt_launder(const char *p) {
    asm volatile("" : "+s"(p));
    return (gchar_p)p;
}
__device__ __forceinline__ f4 ld4g(gfloat_p p) { return *reinterpret_cast<const __attribute__((address_space(1))) f4 *>(p); }
// WT_ABL: timing ablations (results are then meaningless): 1 no GELU, 2 no MFMA, 4 no node-term gathers, 8 no e-tile loads
#ifndef WT_ABL
#define WT_ABL 0
#endif
#if WT_ABL & 2
#define WT_MFMA(a, b, c) (c + __builtin_bit_cast(f4, a) * __builtin_bit_cast(f4, b))
#else
#define WT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0)
#endif
#if WT_ABL & 1
#define WT_GELU4(v) (v)
#else
#define WT_GELU4(v) gelu4(v)
#endif
__device__ __forceinline__ f4 wt_ld_gather(const float *p, int lane) {
#if WT_ABL & 4
    return f4{0.001f * lane, 0.f, 0.1f, 0.2f};
#else
    return ld4(p);
#endif
}
__device__ __forceinline__ f4 wt_ld_tile(const float *p, int lane) {
#if WT_ABL & 8
    return f4{0.001f * lane, 0.3f, 0.1f, 0.2f};
#else
    return ld4(p);
#endif
}

// Builds the A-fragment image of one 128 x 128 weight (row-major fp32, ld floats per row) in LDS / global memory:
// fragment (cb, c), lane (n, q) <- W[wt_row(cb, n)][32 c + 8 q .. +8). dstL == nullptr skips the l plane.
__device__ __forceinline__ void wt_build_frags(const float *__restrict__ W, int ld, char *dstH, char *dstL, int tid, int nt) {
    for (int idx = tid; idx < 2048; idx += nt) {
        const int ln = idx & 63, f = idx >> 6, cb = f >> 2, c = f & 3;
        const float *src = W + (size_t)wt_row(cb, ln & 15) * ld + 32 * c + 8 * (ln >> 4);
        const XFrag x = wt_split8(ld4(src), ld4(src + 4));
        if (dstH) *reinterpret_cast<u4 *>(dstH + f * WT_FRAG_BYTES + ln * 16) = x.h;
        if (dstL) *reinterpret_cast<u4 *>(dstL + f * WT_FRAG_BYTES + ln * 16) = x.l;
    }
}

// LDS addressing of the fragment image: ds_read offsets are 16-bit, so one VGPR base per 64 KB region (two planes) —
// opaque to the compiler, which would otherwise materialise (and spill) one address VGPR per fragment above 64 KB.
struct FragBase { unsigned r[3]; };
__device__ __forceinline__ FragBase wt_frag_base(const char *sW, int lane) {
    FragBase b;
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)sW + (unsigned)lane * 16u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        b.r[k] = base + 65536u * k;
        asm volatile("" : "+v"(b.r[k]));
    }
    return b;
}
// fragment f (= 4 cb + c) of plane p
__device__ __forceinline__ u4 wt_frag(const FragBase &b, int p, int f) {
    return *reinterpret_cast<const __attribute__((address_space(3))) u4 *>(b.r[p >> 1] + (p & 1) * WT_PLANE_BYTES + f * WT_FRAG_BYTES);
}

// acc[rb] += W[block cb] . X[rb]^T over K = 128; the h-plane A fragments come from LDS plane ph, the l-plane ones from
// fetch_l(step) (LDS, or the L2 stream of the edge kernel's third GEMM), step = 4 cb + c
template <int NR, typename FetchL>
__device__ __forceinline__ void wt_gemm_cb(const FragBase &fb, int ph, FetchL &&fetch_l, int cb, const XFrag (&X)[NR][4], f4 (&acc)[NR]) {
#ifndef WT_TWO_ACC
#define WT_TWO_ACC 0
#endif
#if WT_TWO_ACC
    f4 lo[NR];      // the two residual-plane terms accumulate here: consecutive MFMAs never share an accumulator
#pragma unroll
    for (int rb = 0; rb < NR; ++rb) lo[rb] = f4{0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const u4 ah = wt_frag(fb, ph, cb * 4 + c);
        const u4 al = fetch_l(cb * 4 + c);
#if WT_TWO_ACC
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) lo[rb] = WT_MFMA(al, X[rb][c].h, lo[rb]);
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) acc[rb] = WT_MFMA(ah, X[rb][c].h, acc[rb]);
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) lo[rb] = WT_MFMA(ah, X[rb][c].l, lo[rb]);
#else
        // one accumulator: with two column blocks per scheduling region the compiler alternates their MFMA chains
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) acc[rb] = WT_MFMA(al, X[rb][c].h, acc[rb]);
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) acc[rb] = WT_MFMA(ah, X[rb][c].l, acc[rb]);
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) acc[rb] = WT_MFMA(ah, X[rb][c].h, acc[rb]);
#endif
    }
#if WT_TWO_ACC
#pragma unroll
    for (int rb = 0; rb < NR; ++rb) acc[rb] += lo[rb];
#endif
}

#ifndef WT_PIPE
#define WT_PIPE 1
#endif
// One 128 x 128 GEMM of a pass, column block by column block: init(cb, acc) seeds the accumulators (bias / gathered node
// terms), epi(cb, acc) consumes the finished block (GELU, split into the next GEMM's B fragments, ...). WT_PIPE: the
// epilogue of block cb - 1 sits in the same scheduling region as the MFMAs of block cb, so hipcc interleaves the VALU
// work of one block with the matrix work of the next inside the wavefront (on top of the overlap with the SIMD partner).
#ifndef WT_CBG
#define WT_CBG 2       // column blocks per scheduling region: 2 blocks = 4 independent GELU chains beside 24 MFMAs (16-row passes)
#endif
template <int NR, typename FetchL, typename Init, typename Epi>
__device__ __forceinline__ void wt_gemm8(const FragBase &fb, int ph, FetchL &&fetch_l, const XFrag (&X)[NR][4], Init &&init, Epi &&epi) {
    constexpr int G = NR == 1 ? WT_CBG : 1, NG = 8 / G;
    f4 prev[G][NR];
#pragma unroll
    for (int g = 0; g < NG + WT_PIPE; ++g) {
        f4 acc[G][NR];
        if (g < NG) {
#pragma unroll
            for (int k = 0; k < G; ++k) {
                init(g * G + k, acc[k]);
                wt_gemm_cb<NR>(fb, ph, fetch_l, g * G + k, X, acc[k]);
            }
        }
        if (WT_PIPE) {
            if (g > 0) {
#pragma unroll
                for (int k = 0; k < G; ++k) epi((g - 1) * G + k, prev[k]);
            }
            if (g < NG) {
#pragma unroll
                for (int k = 0; k < G; ++k)
#pragma unroll
                    for (int rb = 0; rb < NR; ++rb) prev[k][rb] = acc[k][rb];
            }
        } else {
#pragma unroll
            for (int k = 0; k < G; ++k) epi(g * G + k, acc[k]);
        }
        WT_FENCE();
    }
}

// sum over the 16 lanes of a DPP row (= the 16 tile rows m of one lane quarter q); the total lands in lane m = 15
__device__ __forceinline__ float wt_row_scan(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
    return x;
}
// sum over the 4 lane quarters (lanes l, l ^ 16, l ^ 32, l ^ 48) — every lane gets the total (see row_stats_partial1b)
__device__ __forceinline__ float wt_quad_sum(float x) {
    {
        const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    }
    {
        const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    }
    return x;
}

// Tiles (residues) of this wavefront. Large batches: XCD-aware — workgroup b runs on XCD b % 8, each XCD gets one
// contiguous eighth of the residue axis and the 8 wavefronts of a workgroup take neighbouring residues (their gathers
// hit the same L2 lines). Small batches: consecutive residues go to different CUs first.
template <int NW>
__device__ __forceinline__ TileRange wt_wave_range(int T, int wv) {
    const int G = gridDim.x, b = blockIdx.x;
    if ((G & 7) == 0 && T >= 8 * NW * G) {
        const int x = b & 7, lb = b >> 3;
        const int s = (int)((long long)T * x / 8), e = (int)((long long)T * (x + 1) / 8);
        return TileRange{s + NW * lb + wv, e, (G >> 3) * NW};  // (G / 8) workgroups x NW wavefronts per XCD
    }
    return TileRange{b + G * wv, T, NW * G};
}

// ------------------------------------------------------------------------------------------------
// message pass
// ------------------------------------------------------------------------------------------------
struct MsgArgsW {
    const float *W1e; int ld1;
    const float *W2, *b2, *P;
    const float *hE;
    const int32_t *E_idx;
    const float *mask;
    float *Ssum, *cnt;
    int T;
};

// One wavefront walks its residues in passes of 16 rows (3 per residue). The loop is software-pipelined ACROSS passes:
// the neighbour indices of pass p + 1 are requested before GEMM 1 of pass p, its e rows and gathered node terms
// before GEMM 2 of pass p — every global load has a full GEMM (several microseconds) to land, so no wavefront ever sits
// on an HBM / L2 round trip in front of its MFMAs (the first version of this kernel spent 38 % of its time there).
struct WtPass { int i, r0; };
__device__ __forceinline__ WtPass wt_next_pass(WtPass p, const TileRange &tr) {
    return p.r0 < 32 ? WtPass{p.i, p.r0 + 16} : WtPass{p.i + tr.step, 0};
}

// NW wavefronts per workgroup, one workgroup per CU: 8 -> 2 per SIMD (256 VGPRs), 12 -> 3 per SIMD (168 VGPRs)
template <bool DEC, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void msg_wt_kernel(MsgArgsW a) {
    __shared__ __attribute__((aligned(16))) char sW[4 * WT_PLANE_BYTES];      // W1e.h | W1e.l | W2.h | W2.l
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    wt_build_frags(a.W1e, a.ld1, sW, sW + WT_PLANE_BYTES, tid, 64 * NW);
    wt_build_frags(a.W2, TM_H, sW + 2 * WT_PLANE_BYTES, sW + 3 * WT_PLANE_BYTES, tid, 64 * NW);
    __syncthreads();
    const TileRange tr = wt_wave_range<NW>(a.T, wv);
    if (tr.begin >= tr.end) return;
    const FragBase fb = wt_frag_base(sW, lane);

    auto load_idx = [&](WtPass p) { return a.E_idx[(size_t)p.i * TM_KS + p.r0 + m]; };
    auto row_ptr = [&](WtPass p) { return a.hE + ((size_t)p.i * TM_KS + p.r0 + m) * TM_H + 8 * q; };

    // pipeline fill: pass 0 completely, index of pass 1
    WtPass cur{tr.begin, 0};
    XFrag X[1][4], Y[1][4];
    f4 gj[8];
    float ma;
    {
        const int j0 = load_idx(cur);
        const int jj = j0 < 0 ? cur.i : j0;
        ma = j0 < 0 ? 0.f : (DEC ? 1.f : a.mask[cur.i] * a.mask[jj]);
        f4 raw[8];
        const float *src = row_ptr(cur);
#pragma unroll
        for (int c = 0; c < 8; ++c) raw[c] = wt_ld_tile(src + 32 * (c >> 1) + 4 * (c & 1), lane);
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) gj[cb] = wt_ld_gather(a.P + (size_t)jj * 256 + 128 + wt_col(cb, q), lane);
#pragma unroll
        for (int c = 0; c < 4; ++c) X[0][c] = wt_split8(raw[2 * c], raw[2 * c + 1]);
    }
    WtPass nxt = wt_next_pass(cur, tr);
    bool has_next = nxt.i < tr.end;
    int j0n = load_idx(has_next ? nxt : cur);
    f4 tot[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) tot[cb] = f4{0.f, 0.f, 0.f, 0.f};
    float cnt = 0.f;

    while (true) {
        WT_FENCE();
        const gfloat_p b2 = wt_launder(a.b2);
        const float *Pi = a.P + (size_t)cur.i * 256;
        const float mi = a.mask[cur.i];
        // ---- GEMM 1 (+ node terms riding in the accumulator) -> GELU -> Y planes ----
        {
            f4 g0 = wt_ld_gather(Pi + wt_col(0, q), lane), g0_of[8];
            wt_gemm8<1>(fb, 0, [&](int st) { return wt_frag(fb, 1, st); }, X,
                [&](int cb, f4 (&acc)[1]) {
                    g0_of[cb] = g0;
                    acc[0] = DEC ? gj[cb] : g0 + gj[cb];
                    if (cb < 7) g0 = wt_ld_gather(Pi + wt_col(cb + 1, q), lane);     // own row: L1 / L2 hit, one block ahead
                },
                [&](int cb, const f4 (&acc)[1]) {
                    f4 v = acc[0];
                    if (DEC) v = g0_of[cb] + mi * v;
                    if (cb & 1) wt_put4<1>(Y[0][cb >> 1], WT_GELU4(v));
                    else wt_put4<0>(Y[0][cb >> 1], WT_GELU4(v));
                });
        }
        // ---- requests of the next pass (its index arrived during GEMM 1): e rows, gathered node terms, the index after ----
        const WtPass ld = has_next ? nxt : cur;              // nothing left: re-read this pass (harmless), no divergent code
        const int jjn = j0n < 0 ? ld.i : j0n;
        const float mask_in = a.mask[ld.i], mask_jn = a.mask[jjn];       // unconditional: consumed after GEMM 2
        // gfx9 retires loads in order (one vmcnt): GEMM 2's biases are requested BEFORE the prefetches so that waiting for a
        // bias never waits for the e rows / node terms of the next pass
        f4 bias[8];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) bias[cb] = ld4g(b2 + wt_col(cb, q));
        f4 rawn[8], gjn[8];
        {
            const float *src = row_ptr(ld);
#pragma unroll
            for (int c = 0; c < 8; ++c) rawn[c] = wt_ld_tile(src + 32 * (c >> 1) + 4 * (c & 1), lane);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) gjn[cb] = wt_ld_gather(a.P + (size_t)jjn * 256 + 128 + wt_col(cb, q), lane);
        }
        const WtPass nn = wt_next_pass(ld, tr);
        const bool has_nn = has_next && nn.i < tr.end;
        const int j0nn = load_idx(has_nn ? nn : ld);
        WT_FENCE();
        // ---- GEMM 2 -> GELU -> masked sum over the rows ----
        wt_gemm8<1>(fb, 2, [&](int st) { return wt_frag(fb, 3, st); }, Y,
            [&](int cb, f4 (&acc)[1]) { acc[0] = bias[cb]; },
            [&](int cb, const f4 (&acc)[1]) {
                tot[cb] += WT_GELU4(acc[0]) * ma;        // slots without a neighbour: zero e row, finite node terms, ma = 0
                touch(tot[cb]);
            });
        cnt += ma;
        if (cur.r0 == 32) {                                  // last pass of the residue: sum over the 16 rows of each quarter
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                f4 t;
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = wt_row_scan(tot[cb][k]);
                if (m == 15) st4(a.Ssum + (size_t)cur.i * TM_H + wt_col(cb, q), t);
                tot[cb] = f4{0.f, 0.f, 0.f, 0.f};
            }
            cnt = wt_row_scan(cnt);
            if (lane == 15) a.cnt[cur.i] = cnt;
            cnt = 0.f;
        }
        if (!has_next) break;
        WT_FENCE();
#pragma unroll
        for (int c = 0; c < 4; ++c) X[0][c] = wt_split8(rawn[2 * c], rawn[2 * c + 1]);
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) gj[cb] = gjn[cb];
        ma = j0n < 0 ? 0.f : (DEC ? 1.f : mask_in * mask_jn);
        cur = nxt; nxt = nn; has_next = has_nn; j0n = j0nn;
    }
}

// ------------------------------------------------------------------------------------------------
// encoder edge update:  h_E <- LN3(h_E + W13 gelu(W12 gelu(W11 [h_i | e | h_j] + b11) + b12) + b13)
// LDS: h planes of W11e, W12, W13 and l planes of W11e, W12 = 160 KB exactly; the l plane of W13 streams from L2
// (pre-built fragment image in the packed weight buffer, 32 KB, the same for every wavefront of the chip).
// ------------------------------------------------------------------------------------------------
struct EdgeArgsW {
    const float *W11e, *W12, *b12, *W13, *b13, *g3, *be3, *P;
    const char *W13l;          // global A-fragment image of the l plane of W13
    float *hE;
    const int32_t *E_idx;
    int T;
};

template <int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void enc_edge_wt_kernel(EdgeArgsW a) {
    __shared__ __attribute__((aligned(16))) char sW[5 * WT_PLANE_BYTES];      // W11e.h | W12.h | W13.h | W11e.l | W12.l
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    wt_build_frags(a.W11e, 384, sW, sW + 3 * WT_PLANE_BYTES, tid, 64 * NW);
    wt_build_frags(a.W12, TM_H, sW + WT_PLANE_BYTES, sW + 4 * WT_PLANE_BYTES, tid, 64 * NW);
    wt_build_frags(a.W13, TM_H, sW + 2 * WT_PLANE_BYTES, nullptr, tid, 64 * NW);
    __syncthreads();
    const TileRange tr = wt_wave_range<NW>(a.T, wv);
    if (tr.begin >= tr.end) return;
    const FragBase fb = wt_frag_base(sW, lane);
    const unsigned lo16 = (unsigned)lane * 16u;

    auto load_idx = [&](WtPass p) { return a.E_idx[(size_t)p.i * TM_KS + p.r0 + m]; };
    auto row_ptr = [&](WtPass p) { return a.hE + ((size_t)p.i * TM_KS + p.r0 + m) * TM_H; };

    // pipeline fill: pass 0 completely, index of pass 1 (same cross-pass software pipeline as the message kernel)
    WtPass cur{tr.begin, 0};
    XFrag X[1][4], Y[1][4];
    f4 gj[8];
    bool valid;
    {
        const int j0 = load_idx(cur);
        const int jj = j0 < 0 ? cur.i : j0;
        valid = j0 >= 0;
        f4 raw[8];
        const float *src = row_ptr(cur) + 8 * q;
#pragma unroll
        for (int c = 0; c < 8; ++c) raw[c] = ld4(src + 32 * (c >> 1) + 4 * (c & 1));
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) gj[cb] = wt_ld_gather(a.P + (size_t)jj * 256 + 128 + wt_col(cb, q), lane);
#pragma unroll
        for (int c = 0; c < 4; ++c) X[0][c] = wt_split8(raw[2 * c], raw[2 * c + 1]);
    }
    WtPass nxt = wt_next_pass(cur, tr);
    bool has_next = nxt.i < tr.end;
    int j0n = load_idx(has_next ? nxt : cur);

    while (true) {
        WT_FENCE();
        const gfloat_p b12 = wt_launder(a.b12), b13 = wt_launder(a.b13), g3 = wt_launder(a.g3), be3 = wt_launder(a.be3);
        const gchar_p gl = wt_launder(a.W13l);
        const float *Pi = a.P + (size_t)cur.i * 256;
        float *rowp = row_ptr(cur);
        // ---- GEMM 1: W11e . e + (W11a h_i + b11) + W11c h_j -> GELU -> Y ----
        {
            f4 g0 = wt_ld_gather(Pi + wt_col(0, q), lane);
            wt_gemm8<1>(fb, 0, [&](int st) { return wt_frag(fb, 3, st); }, X,
                [&](int cb, f4 (&acc)[1]) {
                    acc[0] = g0 + gj[cb];
                    if (cb < 7) g0 = wt_ld_gather(Pi + wt_col(cb + 1, q), lane);
                },
                [&](int cb, const f4 (&acc)[1]) {
                    if (cb & 1) wt_put4<1>(Y[0][cb >> 1], WT_GELU4(acc[0]));
                    else wt_put4<0>(Y[0][cb >> 1], WT_GELU4(acc[0]));
                });
        }
        // ---- requests of the next pass: e rows, gathered node terms, the index after ----
        const WtPass ld = has_next ? nxt : cur;
        const int jjn = j0n < 0 ? ld.i : j0n;
        const bool validn = j0n >= 0;
        f4 bias[8];          // GEMM 2's biases first (in-order vmcnt, see the message kernel)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) bias[cb] = ld4g(b12 + wt_col(cb, q));
        f4 rawn[8], gjn[8];
        {
            const float *src = row_ptr(ld) + 8 * q;
#pragma unroll
            for (int c = 0; c < 8; ++c) rawn[c] = ld4(src + 32 * (c >> 1) + 4 * (c & 1));
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) gjn[cb] = wt_ld_gather(a.P + (size_t)jjn * 256 + 128 + wt_col(cb, q), lane);
        }
        const WtPass nn = wt_next_pass(ld, tr);
        const bool has_nn = has_next && nn.i < tr.end;
        const int j0nn = load_idx(has_nn ? nn : ld);
        WT_FENCE();
        // ---- GEMM 2 -> GELU (its output planes reuse X) ----
        wt_gemm8<1>(fb, 1, [&](int st) { return wt_frag(fb, 4, st); }, Y,
            [&](int cb, f4 (&acc)[1]) { acc[0] = bias[cb]; },
            [&](int cb, const f4 (&acc)[1]) {
                if (cb & 1) wt_put4<1>(X[0][cb >> 1], WT_GELU4(acc[0]));
                else wt_put4<0>(X[0][cb >> 1], WT_GELU4(acc[0]));
            });
        // ---- GEMM 3 + residual; the l-plane fragments of W13 stream from L2 three k-steps ahead ----
        f4 v[8];
        {
            constexpr int D = 3;
            u4 alq[D];
#pragma unroll
            for (int st = 0; st < D; ++st) alq[st] = *reinterpret_cast<const __attribute__((address_space(1))) u4 *>(gl + st * WT_FRAG_BYTES + lo16);
            f4 res = ld4(rowp + wt_col(0, q)), res_of[8];
            wt_gemm8<1>(fb, 2,
                [&](int st) {
                    const u4 al = alq[st % D];
                    if (st + D < 32) alq[st % D] = *reinterpret_cast<const __attribute__((address_space(1))) u4 *>(gl + (st + D) * WT_FRAG_BYTES + lo16);
                    return al;
                },
                X,
                [&](int cb, f4 (&acc)[1]) {
                    acc[0] = ld4g(b13 + wt_col(cb, q));
                    res_of[cb] = res;
                    if (cb < 7) res = ld4(rowp + wt_col(cb + 1, q));       // the fp32 residual: re-read from L2 one block ahead
                },
                [&](int cb, const f4 (&acc)[1]) {
                    v[cb] = res_of[cb] + acc[0];
                    touch(v[cb]);
                });
        }
        // ---- LayerNorm 3 (nn.LayerNorm: biased variance, eps 1e-5) over the 128 features of the row: 32 values in this
        // lane, the other 96 in the lanes of the same m — two passes, as the reference does ----
        {
            f4 s4 = v[0];
#pragma unroll
            for (int cb = 1; cb < 8; ++cb) s4 += v[cb];
            const float mean = wt_quad_sum((s4.x + s4.y) + (s4.z + s4.w)) * (1.0f / 128.0f);
            f4 d4 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const f4 d = v[cb] - mean;
                d4 += d * d;
            }
            const float var = wt_quad_sum((d4.x + d4.y) + (d4.z + d4.w)) * (1.0f / 128.0f);
            const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
            WT_FENCE();
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const int col = wt_col(cb, q);
                const f4 y = (v[cb] - mean) * rstd * ld4g(g3 + col) + ld4g(be3 + col);
                // slots without a neighbour keep the zeros the featurizer wrote
                st4(rowp + col, valid ? y : f4{0.f, 0.f, 0.f, 0.f});
                if (cb & 1) WT_FENCE();
            }
        }
        if (!has_next) break;
        WT_FENCE();
#pragma unroll
        for (int c = 0; c < 4; ++c) X[0][c] = wt_split8(rawn[2 * c], rawn[2 * c + 1]);
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) gj[cb] = gjn[cb];
        valid = validn; cur = nxt; nxt = nn; has_next = has_nn; j0n = j0nn;
    }
}

__global__ void wt_prep_lplane_kernel(const float *__restrict__ W, int ld, char *dst) {
    wt_build_frags(W, ld, nullptr, dst, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

}  // namespace

int launch_wt_prep(const float *W13, char *dst_lplane, hipStream_t st) {
    wt_prep_lplane_kernel<<<8, 256, 0, st>>>(W13, TM_H, dst_lplane);
    return tm_check_launch("wt_prep");
}

static int wt_waves() {      // TMPNN_WT_WAVES = 8 | 12 | 16 wavefronts per workgroup
    static const int nw = [] { const char *e = getenv("TMPNN_WT_WAVES"); const int v = e ? atoi(e) : 8; return v == 4 || v == 12 || v == 16 ? v : 8; }();
    return nw;
}
static int wt_grid(int64_t T, int nw) {
    const int64_t cap = tm_num_cus(), need = (T + nw - 1) / nw;
    return (int)(need < cap ? (need < 1 ? 1 : need) : cap);
}

int launch_msg_wt(bool dec, const float *W1e, int ld1, const float *W2, const float *b2, const float *P, const float *hE,
                  const int32_t *E_idx, const float *mask, int64_t T, float *Ssum, float *cnt, hipStream_t st) {
    MsgArgsW a{W1e, ld1, W2, b2, P, hE, E_idx, mask, Ssum, cnt, (int)T};
    const int nw = wt_waves(), grid = wt_grid(T, nw);
#define WT_LAUNCH_MSG(NW)                                                    \
    if (dec) msg_wt_kernel<true, NW><<<grid, 64 * NW, 0, st>>>(a);           \
    else msg_wt_kernel<false, NW><<<grid, 64 * NW, 0, st>>>(a)
    if (nw == 16) { WT_LAUNCH_MSG(16); }
    else if (nw == 4) { WT_LAUNCH_MSG(4); }
    else if (nw == 12) { WT_LAUNCH_MSG(12); }
    else { WT_LAUNCH_MSG(8); }
#undef WT_LAUNCH_MSG
    return tm_check_launch(dec ? "dec_msg_wt" : "enc_msg_wt");
}

int launch_enc_edge_wt(const EncW &e, const char *W13l, const float *P, float *hE, const int32_t *E_idx, int64_t T, hipStream_t st) {
    EdgeArgsW a{e.W11 + 128, e.W12, e.b12, e.W13, e.b13, e.norm3_w, e.norm3_b, P, W13l, hE, E_idx, (int)T};
    const int nw = wt_waves(), grid = wt_grid(T, nw);
    if (nw == 16) enc_edge_wt_kernel<16><<<grid, 1024, 0, st>>>(a);
    else if (nw == 4) enc_edge_wt_kernel<4><<<grid, 256, 0, st>>>(a);
    else if (nw == 12) enc_edge_wt_kernel<12><<<grid, 768, 0, st>>>(a);
    else enc_edge_wt_kernel<8><<<grid, 512, 0, st>>>(a);
    return tm_check_launch("enc_edge_wt");
}
