// Device-side building blocks shared by the ThermoMPNN kernels (gfx950 / CDNA4 only).
//
// Tile convention: every per-edge / per-node GEMM works on a TILE of 48 rows x 128 fp32 features held
// in LDS (one residue's 48 neighbour slots, or 48 residues). A 256-thread workgroup = 4 wavefronts;
// wavefront w owns output columns [32w, 32w+32) as two 16-column blocks and keeps the matching slice
// of the weight matrix in VGPRs. The matrix core instruction is v_mfma_f32_16x16x4_f32 (exact fp32
// FMA chains), used "transposed": the WEIGHT slice is the MFMA A operand (i = output column), the
// ACTIVATION tile is the B operand (j = row), so lane (j = lane&15, q = lane>>4) ends up holding four
// CONSECUTIVE output columns 4q..4q+3 of row j — one 16-byte LDS/global access per 16x16 block.
//
// LDS tiles are row-major with an XOR swizzle on the 16-byte chunk index (chunk ^ (row & 15)): the
// ds_read_b128 of a B fragment (16 rows x same chunk) and the ds_write_b128 of the epilogue are both
// bank-conflict free (MI355X_MICROARCH.md §LDS lane groups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TM_H 128
#define TM_KS 48
#define TM_TILE 48
#define TM_THREADS 256
#ifndef TM_SCHED_FILL
#define TM_SCHED_FILL 4
#endif

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Workgroup / thread indices straight from the hardware registers. The library is built with -mno-amdgpu-ieee, and hipcc then
// refuses to inline the ockl helpers that sit behind blockIdx / threadIdx / gridDim (the callee's FP-mode attributes differ): every
// kernel called them through s_swappc and — worse — treated the returned values as DIVERGENT, so the tile index, every per-tile
// address and every loop bound lived in VGPRs and was recomputed with 64-bit VALU arithmetic (round 3: 3 calls per kernel, 20-30
// VALU instructions per tile in the per-edge kernels). These read the same values the inlined helpers would: SGPRs, and for
// gridDim / blockDim the hidden kernel arguments of code-object v5 (hidden_block_count_x at byte 0, hidden_group_size_x at byte 12
// of the implicit-argument block — device memory, already in the scalar cache with the explicit arguments). NOT
// __builtin_amdgcn_grid_size_x(): that one reads the AQL dispatch packet, which lives in host memory (measured: +0.25 ms on a
// 19-launch single-protein forward).
__device__ __forceinline__ int tm_bid() { return (int)__builtin_amdgcn_workgroup_id_x(); }
__device__ __forceinline__ int tm_tid() { return (int)__builtin_amdgcn_workitem_id_x(); }
// wavefront index inside the (1-D) workgroup as a SCALAR: tid >> 6 is wave-uniform, but only readfirstlane tells the compiler so —
// with it the per-wavefront row / column bases, tile pointers and loop bounds are SALU work and scalar loads
__device__ __forceinline__ int tm_wave(int tid) { return __builtin_amdgcn_readfirstlane(tid >> 6); }
__device__ __forceinline__ int tm_nblk() {
    return (int)((const __attribute__((address_space(4))) unsigned *)__builtin_amdgcn_implicitarg_ptr())[0];
}
__device__ __forceinline__ int tm_bdim() {
    return (int)((const __attribute__((address_space(4))) unsigned short *)__builtin_amdgcn_implicitarg_ptr())[6];
}

// Persistent-kernel work split that is XCD-aware: workgroup b is observed to run on XCD b % 8 (placement is a
// speed hint only, never relied on for correctness). Giving each XCD one CONTIGUOUS eighth of the packed
// residue axis keeps the rows its tiles gather (neighbours live in the same protein) inside that XCD's 4 MB L2
// instead of bouncing the whole [T,256] projection table through every L2 (MI355X_MICROARCH.md: per-XCD L2).
struct TileRange { int begin, end, step; };
__device__ __forceinline__ TileRange xcd_tile_range(int n_tiles) {
    const int G = tm_nblk(), b = tm_bid();
    if ((G & 7) == 0 && n_tiles >= 8 * G) {
        const int x = b & 7, lb = b >> 3;
        const int s = (int)((long long)n_tiles * x / 8), e = (int)((long long)n_tiles * (x + 1) / 8);
        return TileRange{s + lb, e, G >> 3};
    }
    return TileRange{b, n_tiles, G};
}

// float offset of 16-byte chunk `c` of row `m` in a swizzled tile whose rows are RS floats long
template <int RS = 128>
__device__ __forceinline__ int chunk_off(int m, int c) { return m * RS + ((c ^ (m & 15)) << 2); }

// Kernel arguments live in device memory; hipcc fetches them in several dependent groups (SGPR pressure, control flow), each
// a ~0.6 us round trip on a cold argument buffer. One dword of every 64-byte line, requested back to back at kernel entry,
// pulls the whole argument block into the scalar cache in ONE round trip; the compiler's own s_loads then hit.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
    typedef const __attribute__((address_space(4))) unsigned *kptr;
    kptr k = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned any = 0;
#pragma unroll
    for (int o = 0; o < BYTES; o += 64) any |= k[o / 4];
    asm volatile("" ::"s"(any));
}

__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ void st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }

// erf with max abs error 9.3e-8 (< 1 ulp of 1.0f) in ~25 VALU ops, branch-free (both pieces evaluated, one select):
//   |x| <= 1 : x + x P5(x^2)        |x| > 1 : sign(x) (1 - exp(-t Q6(t))), t = min(|x|, 4)
// coefficients fitted and the error measured in emulated fp32-FMA arithmetic by tools/fit_erf.py.
// ocml's erff evaluates two longer branches under divergence and costs about twice as much.
__device__ __forceinline__ float erf_fast(float x) {
    const float t = fminf(fabsf(x), 4.0f), s = x * x;
    float p = -5.654108881e-04f;
    p = fmaf(p, s, 4.923280883e-03f);
    p = fmaf(p, s, -2.671639070e-02f);
    p = fmaf(p, s, 1.128036476e-01f);
    p = fmaf(p, s, -3.761234987e-01f);
    p = fmaf(p, s, 1.283791274e-01f);
    const float small = fmaf(p, x, x);
    float q = 1.617558732e-05f;
    q = fmaf(q, t, -3.671159600e-04f);
    q = fmaf(q, t, 3.792551949e-03f);
    q = fmaf(q, t, -2.399282305e-02f);
    q = fmaf(q, t, 1.063736250e-01f);
    q = fmaf(q, t, 6.351676462e-01f);
    q = fmaf(q, t, 1.128615231e+00f);
    const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(q * t * -1.4426950408889634f), x);
    return t > 1.0f ? big : small;
}
// exact-erf GELU (torch.nn.GELU() default; protein_mpnn_utils.py:813,856,888) in ONE branch-free piece, 16 VALU ops:
//   gelu(x) = x Phi(x),  Phi = 0.5 erfc(-x/sqrt2) = 0.5 + copysign(0.5 - h, x),  h = 0.5 exp(-t Q(t)) = exp2(t Q'(t) - 1),
//   t = min(|x|/sqrt2, 4);  Q' = degree-8 fit of -log2(e) * (-ln erfc(t) / t) on [0, 4]  (tools/fit_gelu.py).
// Max abs error 4.5e-7 over all x (2.7e-7 for |x| < 3) — the fp32 rounding floor of x*Phi itself is 2.4e-7 at |x| = 4.
// In the split-precision kernels GELU is ~80 % of the VALU work, which is what bounds them.
__device__ __forceinline__ float gelu1(float x) {
    const float t = fminf(fabsf(x) * 0.70710678118654752440f, 4.0f);
    float q = 2.814671218e-06f;
    q = fmaf(q, t, -5.093904975e-05f);
    q = fmaf(q, t, 3.626021436e-04f);
    q = fmaf(q, t, -1.058569948e-03f);
    q = fmaf(q, t, -1.620148622e-03f);
    q = fmaf(q, t, 2.903427724e-02f);
    q = fmaf(q, t, -1.488050018e-01f);
    q = fmaf(q, t, -9.183712091e-01f);
    q = fmaf(q, t, -1.627908858e+00f);
    const float h = __builtin_amdgcn_exp2f(fmaf(q, t, -1.0f));
    return x * (0.5f + copysignf(0.5f - h, x));
}
// Two values at a time: the Horner chain, the scalings and the final products run as packed fp32 ops
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes-worth of fp32 per issue slot) — same arithmetic as gelu1.
typedef float f2 __attribute__((ext_vector_type(2)));
// gelu(x) = max(x, 0) - u h(u),  u = min(|x|, 4 sqrt2),  h(u) = 0.5 erfc(u / sqrt2) = exp2(P(u)) (tools/fit_gelu.py). Per value:
// v_min (|x| as a source modifier), 6 Horner steps (packed two values at a time), v_exp, v_max and one fma.
// One packed Horner step q <- q t + c, c broadcast from the low half of an SGPR pair. Written as inline asm because hipcc
// splits about a third of these v_pk_fma_f32 back into two v_fma_f32 when they sit near MFMAs — sensible for an
// MFMA-bound loop, but these kernels are VALU-issue-bound and a packed op costs the same ~4 cycles as a scalar one.
// (Operands are VALU-produced values only, so no MFMA read hazard hides inside the asm.)
__device__ __forceinline__ f2 pk_horner(f2 q, f2 t, float c) {
    f2 r;
    const unsigned long long cc = (unsigned long long)__float_as_uint(c);
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(q), "v"(t), "s"(cc));
    return r;
}
// The exponent polynomial P: degree 6 with a free constant term, fitted to the error of gelu itself (tools/fit_gelu.py: max abs error
// 2.8e-7 — the fp32 rounding floor of x Phi(x) is 2.4e-7) — 6 packed Horner steps. (Rounds 1-2: a degree-8 fit of the exponent with
// the constant pinned to -1, same accuracy, 4 more VALU instructions per pair of values.)
#ifndef TM_GELU_NAN3
#define TM_GELU_NAN3 0     // 1 (the f16x2 per-edge / node files, built WITH NaN semantics: build.py FILE_FLAGS): NaN-propagating clamps, clamped-t tail
#endif
// Timing-only ablations (tools/ablate.sh: ONE ingredient removed, results wrong by construction) exist in debug builds only — the
// shipped library compiles exactly one form of every kernel: TM_ABL_NOGELU (GELU = identity), TM_ABL_NOSPLIT (high plane only),
// TM_ABL_NOMFMA (tile GEMMs issue nothing), TM_ABL_NOLN (edge update without LayerNorm statistics), TM_ABL_NOLOAD (per-edge kernels
// never fetch the next tile), TM_ABL_NOGAUSS (featurizer without Gaussians), TM_ABL_WAVE_NOLDS (wavefront-per-residue message kernel: the
// first two groups' weight fragments reused instead of read from LDS).
#ifndef TMPNN_DEBUG_BUILD
#if defined(TM_ABL_NOGELU) || defined(TM_ABL_NOSPLIT) || defined(TM_ABL_NOMFMA) || defined(TM_ABL_NOLN) || defined(TM_ABL_NOLOAD) || defined(TM_ABL_NOGAUSS) || defined(TM_ABL_WAVE_NOLDS)
#error "TM_ABL_* timing ablations need -DTMPNN_DEBUG_BUILD (python -m thermompnn_amd.build --variant NAME -DTMPNN_DEBUG_BUILD -DTM_ABL_...)"
#endif
#endif
#ifndef TM_ABL_NOGELU
#define TM_ABL_NOGELU 0
#endif
#ifndef TM_ABL_NOSPLIT
#define TM_ABL_NOSPLIT 0
#endif
#ifndef TM_ABL_NOMFMA
#define TM_ABL_NOMFMA 0
#endif
#ifndef TM_ABL_NOLN
#define TM_ABL_NOLN 0
#endif
#ifndef TM_ABL_NOLOAD
#define TM_ABL_NOLOAD 0
#endif
#ifndef TM_ABL_NOGAUSS
#define TM_ABL_NOGAUSS 0
#endif
__device__ __forceinline__ f2 gelu2(f2 x) {
#if TM_ABL_NOGELU
    return x;
#endif
#if TM_GELU_NAN3
    const f2 t = f2{__builtin_elementwise_minimum(fabsf(x.x), 5.656854249f), __builtin_elementwise_minimum(fabsf(x.y), 5.656854249f)};
#else
    const f2 t = f2{fminf(fabsf(x.x), 5.656854249f), fminf(fabsf(x.y), 5.656854249f)};
#endif
    f2 q = __builtin_elementwise_fma(f2{3.309543916e-05f, 3.309543916e-05f}, t, f2{-7.692427171e-04f, -7.692427171e-04f});
    q = pk_horner(q, t, 8.080792133e-03f);
    q = pk_horner(q, t, -5.341222090e-02f);
    q = pk_horner(q, t, -4.587708865e-01f);
    q = pk_horner(q, t, -1.151201730e+00f);
    const f2 e = pk_horner(q, t, -9.999930581e-01f);
    // the library is built with -mno-amdgpu-ieee -fno-honor-nans: fminf / fmaxf are single v_min / v_max (no canonicalising
    // v_max x, x in front of each)
#if TM_GELU_NAN3
    // max(x, 0) - t 2^e with the CLAMPED t as a negated source of ONE packed fma (13 VALU per pair). The clamps are gfx950's
    // v_minimum3_f32 / v_maximum3_f32: a NaN in x — what an f16 overflow always turns into: h = +-inf and l = -+inf meet in one
    // accumulator — comes out as NaN (DESIGN "Range and precision"). Beyond the clamp 2^e = Phi(-5.66) = 7.7e-9, so t and |x| differ
    // by (|x| - 5.66) 7.7e-9 in a result that is 0 or x to that order either way.
    return __builtin_elementwise_fma(-t, f2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)},
                                     f2{__builtin_elementwise_maximum(x.x, 0.f), __builtin_elementwise_maximum(x.y, 0.f)});
#else
    // translation units built with -fno-honor-nans: v_min / v_max return their finite operand, so the last product uses |x| ITSELF —
    // a product with t would launder the NaN of an f16 overflow into a finite value. hipcc packs the two fmas and materialises -|x|
    // with a v_or each (15 VALU per pair); keeping them scalar (abs / neg as free source modifiers) costs more s_nop behind the v_exp
    // than the v_or it saves.
    return f2{fmaf(-fabsf(x.x), __builtin_amdgcn_exp2f(e.x), fmaxf(x.x, 0.f)), fmaf(-fabsf(x.y), __builtin_amdgcn_exp2f(e.y), fmaxf(x.y, 0.f))};
#endif
}
__device__ __forceinline__ f4 gelu4(f4 v) {
    const f2 a = gelu2(f2{v.x, v.y}), b = gelu2(f2{v.z, v.w});
    return f4{a.x, a.y, b.x, b.y};
}

// Weight fragment for one 16-column block: wr[4*kk+s] = W[(n0 + lane&15) * ld + k0 + 16*kk + 4*(lane>>4) + s].
// Rows >= n_rows (ragged last block, e.g. 21 logits) read as zero.
template <int NK16>
__device__ __forceinline__ void load_wfrag(const float *__restrict__ W, int ld, int n0, int k0, int n_rows,
                                           float (&wr)[NK16 * 4], int lane) {
    const int row = n0 + (lane & 15);
    const float *p = W + (size_t)row * ld + k0 + 4 * (lane >> 4);
    const bool ok = row < n_rows;
#pragma unroll
    for (int kk = 0; kk < NK16; ++kk) {
        f4 v = ok ? ld4(p + 16 * kk) : f4{0.f, 0.f, 0.f, 0.f};
        wr[4 * kk + 0] = v.x; wr[4 * kk + 1] = v.y; wr[4 * kk + 2] = v.z; wr[4 * kk + 3] = v.w;
    }
}

// acc[rb][cb] += W_cb . tile^T over K = 16*NK16, for the 3 row blocks of a 48-row tile.
template <int NK16, int NCB, int RS = 128, int NRB = 3>   // NRB = 16-row blocks in the tile (3 = 48 rows)
__device__ __forceinline__ void mma_tile(const float *tile, const float (&w)[NCB][NK16 * 4],
                                         f4 (&acc)[NRB][NCB], int lane) {
    const int m = lane & 15, q = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < NK16; ++kk) {
        f4 a[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) a[rb] = ld4(tile + chunk_off<RS>(16 * rb + m, 4 * kk + q));
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = mfma16(w[cb][4 * kk + s], a[rb][s], acc[rb][cb]);
    }
}

// Same GEMM with a slice of unrelated VALU / LDS / global work issued inside every 16-deep k-step (`slice(kk)`,
// kk a compile-time constant after unrolling). The work in `slice` must be independent of `acc`; the matrix pipe
// and the VALU are separate, so the hardware overlaps the two streams when they are interleaved in program
// order. B fragments are fetched one k-step ahead.
template <int NK16, int NCB, int RS = 128, typename F>
__device__ __forceinline__ void mma_tile_with(const float *tile, const float (&w)[NCB][NK16 * 4], f4 (&acc)[3][NCB],
                                              int lane, F &&slice) {
    const int m = lane & 15, q = lane >> 4;
    f4 a[3];
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) a[rb] = ld4(tile + chunk_off<RS>(16 * rb + m, q));
#pragma unroll
    for (int kk = 0; kk < NK16; ++kk) {
        f4 an[3];
        if (kk + 1 < NK16) {
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) an[rb] = ld4(tile + chunk_off<RS>(16 * rb + m, 4 * (kk + 1) + q));
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = mfma16(w[cb][4 * kk + s], a[rb][s], acc[rb][cb]);
        slice(kk);
        if (kk + 1 < NK16) {
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) a[rb] = an[rb];
        }
#if TM_SCHED_FILL > 0
        // pin an even interleave: after every MFMA at most TM_SCHED_FILL non-matrix instructions (VALU | SALU | VMEM |
        // DS | TRANS). One wavefront per SIMD issues roughly one instruction per 4 cycles, so a 32-cycle
        // v_mfma_f32_16x16x4_f32 hides about five of them (MI355X_MICROARCH.md, per-instruction constants).
#pragma unroll
        for (int g = 0; g < 12 * NCB; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x496, TM_SCHED_FILL, 0);
        }
#endif
    }
}

// Cooperative load of a contiguous [48,128] fp32 block from global into a swizzled LDS tile
// (coalesced 16-byte loads; 6 per thread). rows_valid < 48 zero-fills the tail rows.
template <int NRB = 3>
__device__ __forceinline__ void load_tile(float *tile, const float *__restrict__ src, int rows_valid, int tid) {
#pragma unroll
    for (int it = 0; it < 2 * NRB; ++it) {
        const int idx = it * TM_THREADS + tid;
        const int row = idx >> 5, c = idx & 31;
        f4 v = row < rows_valid ? ld4(src + (size_t)idx * 4) : f4{0.f, 0.f, 0.f, 0.f};
        st4(tile + chunk_off(row, c), v);
    }
}

// Asynchronous variant: LDS-DMA (global_load_lds_dwordx4) straight into a swizzled tile, no VGPR round trip.
// One wave-instruction fills two rows (64 lanes x 16 B, LDS destination = wave-uniform base + lane*16); the
// XOR swizzle is applied on the per-lane SOURCE address (cdna_hip_programming.md rule 21) — a permutation
// inside the row's 512 B, so the global access stays fully coalesced. Each wavefront issues 6 instructions.
// The data is visible after the next __syncthreads() (hipcc drains vmcnt before the barrier).
template <int NW = 4>   // wavefronts per workgroup sharing the 24 row-pair instructions
__device__ __forceinline__ void load_tile_async(float *tile, const float *__restrict__ src, int wv, int lane) {
#pragma unroll
    for (int k = 0; k < 24 / NW; ++k) {
        const int rp = (24 / NW) * wv + k;                // row pair
        const int row = 2 * rp + (lane >> 5);
        const int c = (lane & 31) ^ (row & 15);           // logical chunk that belongs at physical slot (lane & 31)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(src + (size_t)row * TM_H + 4 * c),
            (__attribute__((address_space(3))) void *)(tile + rp * 2 * TM_H), 16, 0, 0);
    }
}

// LDS-DMA issued through inline asm: hipcc treats the builtin above conservatively (s_waitcnt vmcnt(0) in front of the
// first LDS read after it — the copy ends up synchronous). Here the wait is ours to place: lds_dma_wait() before the
// barrier that publishes the data. Hidden VM operations only make the compiler's own vmcnt(N) waits more conservative.
// One call copies 64 x 16 B: lane l's 16 bytes at `g` land at lds_wave_base + 16 l.
__device__ __forceinline__ void lds_dma16(const float *g, const void *lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory", "m0");
}
// Pins the point where a loaded value must have arrived. gfx9 counts loads AND stores in one in-order vmcnt: a value
// first used after later stores were issued makes hipcc wait for those stores too (s_waitcnt vmcnt(N) cannot skip them).
// Touching the value just before the stores are issued moves the wait to where it is free.
__device__ __forceinline__ void touch(f4 &v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float half_wave_sum(float v) {   // sum over the 32 lanes of a half-wavefront
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

// LayerNorm of one 128-wide row held as one f4 per lane of a half-wavefront (nn.LayerNorm, eps 1e-5,
// biased variance). g4/b4 = this lane's 4 gamma/beta values.
// s a + c per component as explicit fmas (see layer_norm_row for why)
__device__ __forceinline__ f4 fma4s(float s, f4 a, f4 c) {
    return f4{__builtin_fmaf(s, a.x, c.x), __builtin_fmaf(s, a.y, c.y), __builtin_fmaf(s, a.z, c.z), __builtin_fmaf(s, a.w, c.w)};
}
// The sum of squares and the affine step are written as explicit fma chains: with -ffp-contract=fast hipcc otherwise picks a
// different mul / fma mix for the same source in different kernels (measured: one ulp between two node_update forms), and
// results must not depend on which form of a kernel a batch size selects.
__device__ __forceinline__ f4 layer_norm_row(f4 v, f4 g4, f4 b4) {
    const float mean = half_wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 128.0f);
    const f4 d = v - mean;
    const float ss = __builtin_fmaf(d.w, d.w, __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x)));
    const float var = half_wave_sum(ss) * (1.0f / 128.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const f4 t = d * rstd;
    return f4{__builtin_fmaf(t.x, g4.x, b4.x), __builtin_fmaf(t.y, g4.y, b4.y), __builtin_fmaf(t.z, g4.z, b4.z), __builtin_fmaf(t.w, g4.w, b4.w)};
}
