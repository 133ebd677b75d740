// Split-precision GEMM cores: fp32-class accuracy on the 16-bit matrix cores.
//
// fp32 MFMA on gfx950 runs at the fp32 vector rate (1/16 of the 16-bit rate). A GEMM operand is therefore kept as a
// few 16-bit "planes" whose (scaled) sum reproduces the fp32 value, and a product is a handful of exact 16-bit x 16-bit
// partial products accumulated in fp32 by v_mfma_f32_16x16x32_{f16,bf16}. Two schemes, one kernel source (policy type):
//
//   SplitH2  (default, "f16x2"):  x = h + l, h = fp16(x), l = fp16(x - h) unscaled (an fp16 subnormal for small x: absolute
//            error <= 2^-25 for |x| < 2, 2^-23 relative above). Product = h h + h l + l h: 3 MFMAs per 32-deep step into one
//            fp32 accumulator; the dropped l l term is 2^-22 relative. Needs |x| < 65504 (activations and weights of this
//            network are O(1)..O(100); an overflow poisons the result and raises TMPNN_STATUS_RANGE, never silently).
//   SplitBF3 ("bf16x3"):          x = h + m + l exactly (3 x 8 bits), six partial products hh, hm, mh, hl, lh, mm:
//            6 MFMAs per step; full fp32 range.
//
// Parity (CPU emulation against the reference goldens, every Linear replaced — tests/test_split_precision_sim.py): hidden states 2.6e-6..3.3e-6 (f16x2),
// 2.0e-6 (bf16x3), 2.2e-6..2.4e-6 for plain fp32 in a different summation order; ddG 3e-6 for all three. A three-term
// bf16 variant (hh, hm, mh) gives 4e-5 and is NOT used.
//
// LDS "plane tile": NP planes x 48 rows x 128 16-bit values (12 KB per plane); within a plane a row is 16 chunks of
// 16 B (8 values), the chunk index XOR-ed with (row & 15): the ds_read_b128 of a B fragment (16 rows, same chunk) is
// conflict-free.
#pragma once
#include <type_traits>

#include "tmpnn_common.h"

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// Tunables of the kernels built on these cores (each measured, see docs/NOTEBOOK.md): B-fragment prefetch distances of the tile GEMMs
// (steps ahead, see mma_tile_split), the node kernel's weight-image ring, the thread whose cycle counter the debug build's phase
// timers read.
#define TM_EDGE_PF 2        // edge update: 0.339 ms at 0, 0.329 at 1, 0.323 at 2, 0.326 at 3-4 (round 2)
#define TM_MSG_PF 3         // message kernels
#define TM_NODE_PF 3        // node update, tall tile
#define TM_NODE_DEEP_D 2    // node update, one 16-row tile per workgroup: fragment images in flight ahead of the GEMM unit being computed. Round 5: 2, not 3
                            // — at 3 the two-projection form was 4 VGPRs over its budget (hipcc spilled a freshly LOADED bias vector: `s_waitcnt
                            // vmcnt(0); scratch_store` in the middle of the prologue's load cluster); 15.0 against 15.6 us per launch, L = 256
#define TM_NODE_DEEP_PF 2   // ... its B-fragment prefetch distance (a 16-row GEMM has 4 steps; 3 would hold all four at once: 8 more VGPRs)
#ifndef TM_PROF_TID
#define TM_PROF_TID 0       // thread of workgroup 0 the TMPNN_*_PROF phase timers read (448 = wavefront 7, lowest issue priority)
#endif
#define SPLIT_PLANE_BYTES (TM_TILE * TM_H * 2)   // one 48 x 128 plane of 16-bit values: 12288 B

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ------------------------------------------------------------------------------------------------
// policies
// ------------------------------------------------------------------------------------------------
struct SplitBF3 {
    static constexpr int NP = 3;
    static constexpr bool EXACT = true;          // join2(split2(x)) == x
    // two fp32 -> NP words of two packed 16-bit values (v_cvt_pk_bf16_f32 rounds and packs both, RNE)
    static __device__ __forceinline__ void split2(f2 x, unsigned (&p)[3]) {
        const bf2 h = __builtin_convertvector(x, bf2);
        const f2 r1 = x - __builtin_convertvector(h, f2);
        const bf2 m = __builtin_convertvector(r1, bf2);
        const f2 r2 = r1 - __builtin_convertvector(m, f2);
        const bf2 l = __builtin_convertvector(r2, bf2);
        p[0] = __builtin_bit_cast(unsigned, h);
        p[1] = __builtin_bit_cast(unsigned, m);
        p[2] = __builtin_bit_cast(unsigned, l);
    }
    static __device__ __forceinline__ f2 join2(const unsigned (&p)[3]) {
        const f2 h = __builtin_convertvector(__builtin_bit_cast(bf2, p[0]), f2);
        const f2 m = __builtin_convertvector(__builtin_bit_cast(bf2, p[1]), f2);
        const f2 l = __builtin_convertvector(__builtin_bit_cast(bf2, p[2]), f2);
        return (h + m) + l;
    }
    // acc / lo += W-fragment . X-fragment over one 32-deep step (low-order terms in `lo`)
    static __device__ __forceinline__ void mma(const u4 (&w)[3], const u4 (&x)[3], f4 &acc, f4 &lo) {
#define TM_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0)
        lo = TM_BF(w[2], x[0], lo);    // l h
        lo = TM_BF(w[0], x[2], lo);    // h l
        lo = TM_BF(w[1], x[1], lo);    // m m
        lo = TM_BF(w[1], x[0], lo);    // m h
        lo = TM_BF(w[0], x[1], lo);    // h m
        acc = TM_BF(w[0], x[0], acc);  // h h
#undef TM_BF
    }
    static __device__ __forceinline__ f4 fold(f4 acc, f4 lo) { return acc + lo; }
};

struct SplitH2 {
    static constexpr int NP = 2;
    static constexpr bool EXACT = false;         // |x - (h + l)| <= 2^-25 for |x| < 2, 2^-23 relative above
    // h = fp16(x), l = fp16(x - h) UNSCALED: l may be an fp16 subnormal (the matrix core does not flush them), which bounds
    // the representation error by the fp16 subnormal quantum 2^-24 instead of by 11 more mantissa bits — as good as the
    // scaled residual for the O(1) activations / O(0.1) weights of this network, and it needs no 2^11 scaling here, no
    // second accumulator and no fold after the GEMM (VALU is what bounds these kernels).
    static __device__ __forceinline__ void split2(f2 x, unsigned (&p)[2]) {
        const h2 h = __builtin_convertvector(x, h2);                        // v_cvt_pk_f16_f32, RNE
        p[0] = __builtin_bit_cast(unsigned, h);
#if TM_ABL_NOSPLIT
        p[1] = 0u;
        return;
#endif
        const f2 r = x - __builtin_convertvector(h, f2);                    // exact
        const h2 l = __builtin_convertvector(r, h2);
        p[1] = __builtin_bit_cast(unsigned, l);
    }
    static __device__ __forceinline__ f2 join2(const unsigned (&p)[2]) {
        const f2 h = __builtin_convertvector(__builtin_bit_cast(h2, p[0]), f2);
        const f2 l = __builtin_convertvector(__builtin_bit_cast(h2, p[1]), f2);
        return h + l;
    }
    // all three partial products go into the ONE accumulator (`lo` stays untouched and folds as a no-op)
    static __device__ __forceinline__ void mma(const u4 (&w)[2], const u4 (&x)[2], f4 &acc, f4 &lo) {
#define TM_HF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0)
        acc = TM_HF(w[1], x[0], acc);  // l h
        acc = TM_HF(w[0], x[1], acc);  // h l
        acc = TM_HF(w[0], x[0], acc);  // h h
#undef TM_HF
    }
    static __device__ __forceinline__ f4 fold(f4 acc, f4 lo) { return acc; }
};

// ------------------------------------------------------------------------------------------------
// plane tiles in LDS
// ------------------------------------------------------------------------------------------------
// byte offset inside a plane tile of columns [4*c4, 4*c4+4) of row `row`, plane p. ROWB = bytes per plane row
// (256 for 128-column tiles), ROWS = rows per plane.
template <int ROWS = TM_TILE, int ROWB = 256>
__device__ __forceinline__ int plane_off4(int p, int row, int c4) {
    return p * (ROWS * ROWB) + row * ROWB + ((((c4 >> 1) ^ (row & 15)) << 4) | ((c4 & 1) << 3));
}
// byte offset of the 16-byte chunk c16 (columns [8*c16, 8*c16+8)) of row `row`, plane p
// (SWZ = false: plain row-major planes whose row pitch ROWB is conflict-free by itself, e.g. 848 B = 20 banks mod 64)
template <int ROWS = TM_TILE, int ROWB = 256, bool SWZ = true>
__device__ __forceinline__ int plane_off8(int p, int row, int c16) {
    return p * (ROWS * ROWB) + row * ROWB + ((SWZ ? (c16 ^ (row & 15)) : c16) << 4);
}

// write four consecutive fp32 columns of one row into the planes (one 8-byte packet per plane)
template <typename SP, int ROWS = TM_TILE, int ROWB = 256>
__device__ __forceinline__ void store_split(char *tile, int row, int c4, f4 v) {
    unsigned a[SP::NP], b[SP::NP];
    SP::split2(f2{v.x, v.y}, a);
    SP::split2(f2{v.z, v.w}, b);
#pragma unroll
    for (int p = 0; p < SP::NP; ++p) *reinterpret_cast<u2 *>(tile + plane_off4<ROWS, ROWB>(p, row, c4)) = u2{a[p], b[p]};
}
// reconstruction of four consecutive columns (exact for SplitBF3)
template <typename SP, int ROWS = TM_TILE, int ROWB = 256>
__device__ __forceinline__ f4 load_joined(const char *tile, int row, int c4) {
    unsigned a[SP::NP], b[SP::NP];
#pragma unroll
    for (int p = 0; p < SP::NP; ++p) {
        const u2 w = *reinterpret_cast<const u2 *>(tile + plane_off4<ROWS, ROWB>(p, row, c4));
        a[p] = w.x;
        b[p] = w.y;
    }
    const f2 lo = SP::join2(a), hi = SP::join2(b);
    return f4{lo.x, lo.y, hi.x, hi.y};
}

// ------------------------------------------------------------------------------------------------
// register-resident weight fragments + the tile GEMM
// ------------------------------------------------------------------------------------------------
// Weight fragments of one 16-column block for K = 32*NK32: wf[c].p[plane] = 8 values of
//   W[(n0 + lane&15) * ld + k0 + 32 c + 8 (lane>>4) + j], j = 0..7.   Columns >= k_valid read as zero (K padding), except the
//   first k_wrap of them, which read the SAME row one `ld` back (W[row, k0 + k - ld]: the featurizer's positional columns).
template <typename SP>
struct WFragS { u4 p[SP::NP]; };

// PERM (full 128-deep weights only): the K order of the message kernels (perm_c4 below): lane group q holds k = 32 c + 4 q + {0..3} and
// 32 c + 16 + 4 q + {0..3} of step c.
template <typename SP, int NK32, bool PERM = false>
__device__ __forceinline__ void load_wfrag_split(const float *__restrict__ W, int ld, int n0, int k0, int k_valid,
                                                 WFragS<SP> (&wf)[NK32], int lane, int k_wrap = 0) {
    const float *src = W + (size_t)(n0 + (lane & 15)) * ld + k0 + (PERM ? 4 : 8) * (lane >> 4);
#pragma unroll
    for (int c = 0; c < NK32; ++c) {
        unsigned w[4][SP::NP];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k = 32 * c + (PERM ? 4 : 8) * (lane >> 4) + (PERM ? 16 : 4) * half;
            const f4 v = k < k_valid ? ld4(src + 32 * c + (PERM ? 16 : 4) * half) : k < k_valid + k_wrap ? ld4(src + 32 * c + 4 * half - ld) : f4{0.f, 0.f, 0.f, 0.f};
            SP::split2(f2{v.x, v.y}, w[2 * half]);
            SP::split2(f2{v.z, v.w}, w[2 * half + 1]);
        }
#pragma unroll
        for (int p = 0; p < SP::NP; ++p) wf[c].p[p] = u4{w[0][p], w[1][p], w[2][p], w[3][p]};
    }
}

// acc[rb][cb] += W_cb . tile^T over K = 32*NK32. The low-order terms go through a second accumulator that is folded
// in at the end, so they are not swamped while the leading term is still growing. The weight fragments used are
// w[cb][C0 .. C0+NK32) (a K sub-range of a wider weight); the tile starts at its column 0.
// PF > 0: explicit software pipeline — the B fragments of step s + PF (a step = one row block of one 32-deep chunk) are
// requested before the MFMAs of step s, pinned with scheduling barriers; PF < 0: all row blocks of a chunk requested
// together. Left to itself hipcc, under register pressure, alternates one ds_read with the MFMAs that consume it: the
// featurizer's 39-step GEMM runs at LDS latency (7.4k cycles against 3.7k of matrix-pipe time). Both forms fix the
// schedule (6-7 reads in flight) but need 24-32 more VGPRs than that kernel has (81-94 spilled) — not used there yet.
template <typename SP, int NK32, int NCB, int NRB = 3, int ROWS = TM_TILE, int ROWB = 256, int NKTOT = NK32, int C0 = 0,
          bool SWZ = true, int PF = 0>
__device__ __forceinline__ void mma_tile_split(const char *tile, const WFragS<SP> (&w)[NCB][NKTOT], f4 (&acc)[NRB][NCB], int lane) {
#if TM_ABL_NOMFMA
    return;
#endif
    const int m = lane & 15, q = lane >> 4;
    f4 lo[NRB][NCB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) lo[rb][cb] = f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PF > 0) {
        constexpr int NS = NK32 * NRB, NB = PF + 1;
        u4 x[NB][SP::NP];
#pragma unroll
        for (int s = 0; s < PF && s < NS; ++s)
#pragma unroll
            for (int p = 0; p < SP::NP; ++p)
                x[s][p] = *reinterpret_cast<const u4 *>(tile + plane_off8<ROWS, ROWB, SWZ>(p, 16 * (s % NRB) + m, 4 * (s / NRB) + q));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + PF < NS) {
                const int sn = s + PF;
#pragma unroll
                for (int p = 0; p < SP::NP; ++p)
                    x[sn % NB][p] = *reinterpret_cast<const u4 *>(tile + plane_off8<ROWS, ROWB, SWZ>(p, 16 * (sn % NRB) + m, 4 * (sn / NRB) + q));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) SP::mma(w[cb][C0 + s / NRB].p, x[s % NB], acc[s % NRB][cb], lo[s % NRB][cb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (PF < 0) {
        // grouped: all NRB row blocks' fragments of a chunk are requested together, then its NRB x NCB x terms MFMAs run
        // (one LDS latency per chunk instead of one per row block; the SIMD's other wavefront fills the gap)
#pragma unroll
        for (int c = 0; c < NK32; ++c) {
            u4 x[NRB][SP::NP];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int p = 0; p < SP::NP; ++p)
                    x[rb][p] = *reinterpret_cast<const u4 *>(tile + plane_off8<ROWS, ROWB, SWZ>(p, 16 * rb + m, 4 * c + q));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) SP::mma(w[cb][C0 + c].p, x[rb], acc[rb][cb], lo[rb][cb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int c = 0; c < NK32; ++c) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                u4 x[SP::NP];                               // one row block at a time: few B-fragment VGPRs in flight
#pragma unroll
                for (int p = 0; p < SP::NP; ++p)
                    x[p] = *reinterpret_cast<const u4 *>(tile + plane_off8<ROWS, ROWB, SWZ>(p, 16 * rb + m, 4 * c + q));
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) SP::mma(w[cb][C0 + c].p, x, acc[rb][cb], lo[rb][cb]);
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = SP::fold(acc[rb][cb], lo[rb][cb]);
}

// The PF > 0 form with a rider: ride(S), S = 0 .. NK32 * NRB - 1, is what the caller wants issued behind the MFMAs of step S — global
// requests, one per step (round 6: several global_loads in a row cost their wavefront ~85 cycles of issue each; behind a step's MFMAs
// they cost nothing). Same MFMA order as mma_tile_split: the same bits.
template <typename SP, int NK32, int NRB, int PF, int ROWS = TM_TILE, typename R>
__device__ __forceinline__ void mma_tile_split_ride(const char *tile, const WFragS<SP> (&w)[1][NK32], f4 (&acc)[NRB][1], int lane, R &&ride) {
    constexpr int NS = NK32 * NRB, NB = PF + 1;
#if TM_ABL_NOMFMA
    static_for<0, NS>([&](auto S) { ride(S); });
    return;
#endif
    const int m = lane & 15, q = lane >> 4;
    f4 lo[NRB][1];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) lo[rb][0] = f4{0.f, 0.f, 0.f, 0.f};
    u4 x[NB][SP::NP];
#pragma unroll
    for (int s = 0; s < PF && s < NS; ++s)
#pragma unroll
        for (int p = 0; p < SP::NP; ++p)
            x[s][p] = *reinterpret_cast<const u4 *>(tile + plane_off8<ROWS, 256, true>(p, 16 * (s % NRB) + m, 4 * (s / NRB) + q));
    static_for<0, NS>([&](auto S) {
        constexpr int s = decltype(S)::value;
        if constexpr (s + PF < NS) {
            constexpr int sn = s + PF;
#pragma unroll
            for (int p = 0; p < SP::NP; ++p)
                x[sn % NB][p] = *reinterpret_cast<const u4 *>(tile + plane_off8<ROWS, 256, true>(p, 16 * (sn % NRB) + m, 4 * (sn / NRB) + q));
        }
        __builtin_amdgcn_sched_barrier(0);
        SP::mma(w[0][s / NRB].p, x[s % NB], acc[s % NRB][0], lo[s % NRB][0]);
        __builtin_amdgcn_sched_barrier(0);
        ride(S);
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = SP::fold(acc[rb][0], lo[rb][0]);
}

// One 16-row block at a time (keeps only NCB accumulator pairs live): acc[cb] += W_cb . tile[16 rb .. 16 rb + 16)^T.
// (Interleaving the MFMA chains term by term across accumulators was measured: no gain in the kernels, -12 % in the
// GEMM probe — back-to-back MFMAs on one accumulator do not stall on gfx950.)
template <typename SP, int NK32, int NCB, int ROWS = TM_TILE, int ROWB = 256>
__device__ __forceinline__ void mma_rb_split(const char *tile, int rb, const WFragS<SP> (&w)[NCB][NK32], f4 (&acc)[NCB], int lane) {
    const int m = lane & 15, q = lane >> 4;
    f4 lo[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) lo[cb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NK32; ++c) {
        u4 x[SP::NP];
#pragma unroll
        for (int p = 0; p < SP::NP; ++p)
            x[p] = *reinterpret_cast<const u4 *>(tile + plane_off8<ROWS, ROWB>(p, 16 * rb + m, 4 * c + q));
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) SP::mma(w[cb][c].p, x, acc[cb], lo[cb]);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = SP::fold(acc[cb], lo[cb]);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm statistics fused into a GEMM epilogue (8 wavefronts x 16 columns): each wavefront reduces its 16 columns
// of a row to (mean, M2) over the 4 lane quarters (Chan merge), the 8 partials per row meet in LDS [row][16].
// ------------------------------------------------------------------------------------------------
// The merge partners (lanes l ^ 16, then l ^ 32) are fetched with gfx950's v_permlane16_swap / v_permlane32_swap (VALU,
// no LDS-crossbar round trip): swapping a register with itself leaves {even rows, even rows} in one result and {odd
// rows, odd rows} in the other, and the merge is symmetric, so every lane gets the pair's result without a select.
__device__ __forceinline__ void row_stats_partial1b(const f4 v, float *stat_slot, int q) {
    float mean = (v.x + v.y + v.z + v.w) * 0.25f;
    const f4 d = v - mean;
    float m2 = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    {   // lanes l and l ^ 16: 4 + 4 values
        const auto pm = __builtin_amdgcn_permlane16_swap(__float_as_uint(mean), __float_as_uint(mean), false, false);
        const auto p2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m2), __float_as_uint(m2), false, false);
        const float ma = __uint_as_float(pm[0]), mb = __uint_as_float(pm[1]);
        const float delta = mb - ma;
        mean = 0.5f * (ma + mb);
        m2 = __uint_as_float(p2[0]) + __uint_as_float(p2[1]) + delta * delta * 2.0f;
    }
    {   // lanes l and l ^ 32: 8 + 8 values
        const auto pm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mean), __float_as_uint(mean), false, false);
        const auto p2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m2), __float_as_uint(m2), false, false);
        const float ma = __uint_as_float(pm[0]), mb = __uint_as_float(pm[1]);
        const float delta = mb - ma;
        mean = 0.5f * (ma + mb);
        m2 = __uint_as_float(p2[0]) + __uint_as_float(p2[1]) + delta * delta * 4.0f;
    }
    if (q == 0) *reinterpret_cast<f2 *>(stat_slot) = f2{mean, m2};      // one 8-byte write (stat_slot is 8-byte aligned: even column)
}
// Pitch of the statistics rows these two functions use: 20 floats, not 16. With 16 the 16 lanes of a q = 0 group wrote to two bank
// pairs (8-way conflict) and the 16-byte reads of finish8b below hit every fourth row on the same banks (4-way): the featurizer's
// SQ_LDS_BANK_CONFLICT was 24 % of its LDS cycles (profiles/r03_pmc_sq_summary.txt). 20 keeps the rows 16-byte aligned, makes the
// reads conflict-free (chunk 5 m + k mod 16 is a bijection in m) and leaves the writes 2-way.
#define TM_STAT8_LD 20
__device__ __forceinline__ void row_stats_finish8b(const float *stat_row, float &mean, float &rstd) {
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
    rstd = __builtin_amdgcn_rsqf((m2 + 16.f * dd) * (1.0f / 128.0f) + 1e-5f);     // v_rsq_f32, 1 ulp
}

// ------------------------------------------------------------------------------------------------
// Cheaper form of the same fused statistics (edge update, f16x2): 20 + 11 VALU operations per row instead of 28 + 33.
//  * partial: the wavefront's 16 columns of a row (4 lanes x 4 values) as an exact two-pass — sum over the 16 values (plain adds,
//    permlane swaps), mean, squared deviations from THAT mean, summed the same way — no Chan merge inside the wavefront;
//  * finish: the 8 partials of a row are spread over 8 lanes of the half-wavefront that owns the row in the row phase (lane k
//    loads partial k with one ds_read_b64) and merged with two 8-lane DPP all-reduces (quad_perm xor 1, xor 2, row_half_mirror)
//    instead of every lane reading and merging all 8 partials by itself;
//  * the statistics rows are TM_STAT_LD = 18 floats apart: the 16 lanes of a q = 0 group write 8 bytes each at bank
//    18 m + 2 wv (mod 32) — all distinct (a 16-float pitch put them on two bank pairs: 8-way conflicts).
// Deterministic per row (fixed merge order); not bit-identical to row_stats_partial1b / finish8b, which the bf16x3 kernel keeps.
// ------------------------------------------------------------------------------------------------
#define TM_STAT_LD 18
__device__ __forceinline__ float swap_add16(float x) {      // x(lane) + x(lane ^ 16) on every lane
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
__device__ __forceinline__ float swap_add32(float x) {      // x(lane) + x(lane ^ 32)
    const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
__device__ __forceinline__ void row_stats_partial16(const f4 v, float *stat_slot, int q) {
    const float s = swap_add32(swap_add16((v.x + v.y) + (v.z + v.w)));
    const float mean = s * 0.0625f;
    const f4 d = v - mean;
    const float m2 = swap_add32(swap_add16(__builtin_fmaf(d.w, d.w, __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x)))));
    if (q == 0) *reinterpret_cast<f2 *>(stat_slot) = f2{mean, m2};
}
__device__ __forceinline__ float allreduce8_dpp(float x) {  // sum over each aligned group of 8 lanes, on every lane
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true));    // row_half_mirror
    return x;
}
// stat_row = &s_stat[row][0]; every lane of the half-wavefront that owns `row` calls this
__device__ __forceinline__ void row_stats_finish8d(const float *stat_row, int lane, float &mean, float &rstd) {
    const f2 p = *reinterpret_cast<const f2 *>(stat_row + 2 * (lane & 7));
    mean = allreduce8_dpp(p.x) * 0.125f;
    const float d = p.x - mean;
    const float m2 = allreduce8_dpp(__builtin_fmaf(16.f * d, d, p.y));
    rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(m2, 1.0f / 128.0f, 1e-5f));
}

// ------------------------------------------------------------------------------------------------
// shared by the per-edge kernel files (tmpnn_edge.hip, tmpnn_msg.hip, tmpnn_edge_msg.hip, tmpnn_node.hip)
// ------------------------------------------------------------------------------------------------
// Weight fragment of wavefront wv from a pre-built image (WImg, tmpnn_internal.h): 8 coalesced 16-byte loads instead of the
// 16-row fp32 gathers + on-the-fly split of load_wfrag_split. img == nullptr -> the gather path.
// The message pass's K order (round 5). Its wavefront-per-residue form (msg8_wave_kernel) keeps the activations in the MFMA accumulator
// layout, so lane group q of 32-deep step c holds k = 32 c + 4 q + {0..3} and 32 c + 16 + 4 q + {0..3}. The products of a step are summed
// inside the matrix core in slot order: for a protein's numbers not to depend on which form its launch took, EVERY f16x2 form of the
// message pass uses that order — the 8-wavefront forms write their plane tiles with the column group c4 = 8 c + 4 h + q stored where
// 8 c + 2 q + h would be (a 16-byte chunk then holds exactly one lane group's 8 values), and all read the K-permuted weight images.
__device__ __forceinline__ int perm_c4(int c4) { return (c4 & ~7) | ((c4 & 3) << 1) | ((c4 >> 2) & 1); }

template <typename SP, bool PERM = false>
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
        load_wfrag_split<SP, 4, PERM>(W, ld, 16 * wv, 0, TM_H, wf, lane);
    }
}

struct EdgeArgsB {
    const float *W11e, *W12, *b12, *W13, *b13, *g3, *be3, *P;
    float *hE;
    const int32_t *E_idx;
    int T;
    const char *img11, *img12, *img13;      // fragment images of the three weights (f16x2 only) or null
    const char *imgp11, *imgp12, *imgp13;   // ... their K-permuted forms (perm_c4) or null
};

struct MsgArgsB {
    const float *W1e; int ld1;
    const float *W2, *b2, *P;
    const float *hE;
    const int32_t *E_idx;
    const float *mask;
    float *Ssum, *cnt;
    int T;
    const char *img1, *img2;                // fragment images of W1e / W2 (f16x2 only) or null
    const char *imgp1, *imgp2;              // ... their K-permuted forms (every f16x2 form of the message pass: perm_c4)
    int i0;                                 // first residue of this launch (a large launch = wavefront-per-residue part + a remainder)
};

// accumulator blocks 2 c, 2 c + 1 (fp32, accumulator layout) -> the f16x2 B operand of step c of the next GEMM (K order: perm_c4)
__device__ __forceinline__ void split_pair(const f4 a, const f4 b, u4 (&x)[2]) {
    unsigned p0[2], p1[2], p2[2], p3[2];
    SplitH2::split2(f2{a.x, a.y}, p0);
    SplitH2::split2(f2{a.z, a.w}, p1);
    SplitH2::split2(f2{b.x, b.y}, p2);
    SplitH2::split2(f2{b.z, b.w}, p3);
    x[0] = u4{p0[0], p1[0], p2[0], p3[0]};
    x[1] = u4{p0[1], p1[1], p2[1], p3[1]};
}


