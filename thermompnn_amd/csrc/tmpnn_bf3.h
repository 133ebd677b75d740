// Split-precision GEMM core: fp32-class accuracy on the bf16 matrix cores.
//
// An fp32 value is the EXACT sum of three bf16 values x = h + m + l (8 + 8 + 8 mantissa bits). A product of two such
// numbers is approximated by the six partial products of weight <= 2^-16 (hh, hm, mh, hl, lh, mm; the three dropped
// ones are ~2^-24 relative, i.e. fp32 rounding class). Each partial product is one v_mfma_f32_16x16x32_bf16 (exact
// bf16 x bf16 products, fp32 accumulation), so a 32-deep step costs 6 bf16 MFMAs (~17 cycles each) instead of 8 fp32
// MFMAs (32 cycles each): 2.5x less matrix-pipe time — and the bf16 matrix core is a separate unit from the VALU.
// Parity: emulated on the CPU against the reference goldens (tools/bf3_sim.py): hidden states 2e-6, ddG 3e-6 — the
// same as the exact-fp32 path; the three-term variant (hh, hm, mh) would NOT pass (4e-5).
//
// LDS "plane tile": 3 planes x 48 rows x 128 bf16 (36 KB); within a plane a row is 16 chunks of 16 B (8 bf16), the
// chunk index XOR-ed with (row & 15): the ds_read_b128 of a B fragment (16 rows, same chunk) is conflict-free.
#pragma once
#include "tmpnn_common.h"

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define BF3_PLANE (TM_TILE * TM_H)           // bf16 elements per plane of a 48 x 128 tile
#define BF3_TILE_BYTES (3 * BF3_PLANE * 2)   // 36864

__device__ __forceinline__ f4 mfma_bf16(bf8 a, bf8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// round-to-nearest-even fp32 -> bf16 bits (inputs are finite)
__device__ __forceinline__ unsigned bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_to_f32(unsigned b) { return __uint_as_float(b << 16); }

// x = h + m + l, each a bf16 (returned as 16-bit patterns)
__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    h = bf16_rne(x);
    const float r1 = x - bf16_to_f32(h);
    m = bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(m);
    l = bf16_rne(r2);
}

// four consecutive columns -> three 8-byte packets (one per plane). Written on 2-vectors so that hipcc emits
// v_cvt_pk_bf16_f32 (one instruction rounds and packs two values, RNE) and packed subtracts: ~18 VALU per f4.
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_f2(f2 x, unsigned &ph, unsigned &pm, unsigned &pl) {
    const bf2 h = __builtin_convertvector(x, bf2);
    const f2 r1 = x - __builtin_convertvector(h, f2);
    const bf2 m = __builtin_convertvector(r1, bf2);
    const f2 r2 = r1 - __builtin_convertvector(m, f2);
    const bf2 l = __builtin_convertvector(r2, bf2);
    ph = __builtin_bit_cast(unsigned, h);
    pm = __builtin_bit_cast(unsigned, m);
    pl = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split3_f4(f4 v, u2 &ph, u2 &pm, u2 &pl) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3_f2(f2{v.x, v.y}, h0, m0, l0);
    split3_f2(f2{v.z, v.w}, h1, m1, l1);
    ph = u2{h0, h1};
    pm = u2{m0, m1};
    pl = u2{l0, l1};
}

// byte offset inside a plane tile of columns [4*c4, 4*c4+4) of row `row`, plane p. ROWB = bytes per plane row
// (256 for the 128-column tiles, 512 for the featurizer's half-width RBF tiles), ROWS = rows per plane.
template <int ROWS = TM_TILE, int ROWB = 256>
__device__ __forceinline__ int plane_off4(int p, int row, int c4) {
    return p * (ROWS * ROWB) + row * ROWB + ((((c4 >> 1) ^ (row & 15)) << 4) | ((c4 & 1) << 3));
}
// byte offset of the 16-byte chunk c16 (columns [8*c16, 8*c16+8)) of row `row`, plane p
template <int ROWS = TM_TILE, int ROWB = 256>
__device__ __forceinline__ int plane_off8(int p, int row, int c16) {
    return p * (ROWS * ROWB) + row * ROWB + ((c16 ^ (row & 15)) << 4);
}

// write four consecutive fp32 columns of one row into the three planes
template <int ROWS = TM_TILE, int ROWB = 256>
__device__ __forceinline__ void store_split(char *tile, int row, int c4, f4 v) {
    u2 ph, pm, pl;
    split3_f4(v, ph, pm, pl);
    *reinterpret_cast<u2 *>(tile + plane_off4<ROWS, ROWB>(0, row, c4)) = ph;
    *reinterpret_cast<u2 *>(tile + plane_off4<ROWS, ROWB>(1, row, c4)) = pm;
    *reinterpret_cast<u2 *>(tile + plane_off4<ROWS, ROWB>(2, row, c4)) = pl;
}
// exact reconstruction x = h + m + l of four consecutive columns
template <int ROWS = TM_TILE, int ROWB = 256>
__device__ __forceinline__ f4 load_joined(const char *tile, int row, int c4) {
    const u2 ph = *reinterpret_cast<const u2 *>(tile + plane_off4<ROWS, ROWB>(0, row, c4));
    const u2 pm = *reinterpret_cast<const u2 *>(tile + plane_off4<ROWS, ROWB>(1, row, c4));
    const u2 pl = *reinterpret_cast<const u2 *>(tile + plane_off4<ROWS, ROWB>(2, row, c4));
    f4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned sh = (k & 1) * 16;
        const float h = bf16_to_f32((ph[k >> 1] >> sh) & 0xffffu), m = bf16_to_f32((pm[k >> 1] >> sh) & 0xffffu),
                    l = bf16_to_f32((pl[k >> 1] >> sh) & 0xffffu);
        v[k] = (h + m) + l;
    }
    return v;
}

// Weight fragments of one 16-column block for K = 32*NK32: wf[c][p] = 8 bf16 of plane p:
//   W[(n0 + lane&15) * ld + k0 + 32 c + 8 (lane>>4) + j], j = 0..7.   Columns >= k_valid read as zero (K padding).
struct WFrag3 { bf8 p[3]; };
template <int NK32>
__device__ __forceinline__ void load_wfrag_bf3(const float *__restrict__ W, int ld, int n0, int k0, int k_valid,
                                               WFrag3 (&wf)[NK32], int lane) {
    const float *src = W + (size_t)(n0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4);
#pragma unroll
    for (int c = 0; c < NK32; ++c) {
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k = 32 * c + 8 * (lane >> 4) + 4 * half;
            const f4 v = k < k_valid ? ld4(src + 32 * c + 4 * half) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) split3(v[j], h[4 * half + j], m[4 * half + j], l[4 * half + j]);
        }
        u4 ph, pm, pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ph[j] = h[2 * j] | (h[2 * j + 1] << 16);
            pm[j] = m[2 * j] | (m[2 * j + 1] << 16);
            pl[j] = l[2 * j] | (l[2 * j + 1] << 16);
        }
        wf[c].p[0] = __builtin_bit_cast(bf8, ph);
        wf[c].p[1] = __builtin_bit_cast(bf8, pm);
        wf[c].p[2] = __builtin_bit_cast(bf8, pl);
    }
}

// acc[rb][cb] += W_cb . tile^T over K = 32*NK32 with the six-term split product. The low-order terms go through a
// second accumulator that is folded in at the end, so they are not swamped while the leading term is still growing.
// The weight fragments used are w[cb][C0 .. C0+NK32) (a K sub-range of a wider weight); the tile starts at its column 0.
template <int NK32, int NCB, int NRB = 3, int ROWS = TM_TILE, int ROWB = 256, int NKTOT = NK32, int C0 = 0>
__device__ __forceinline__ void mma_tile_bf3(const char *tile, const WFrag3 (&w)[NCB][NKTOT], f4 (&acc)[NRB][NCB], int lane) {
    const int m = lane & 15, q = lane >> 4;
    f4 lo[NRB][NCB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) lo[rb][cb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NK32; ++c) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            bf8 x[3];                                  // one row block at a time: 12 VGPRs of B fragments in flight
#pragma unroll
            for (int p = 0; p < 3; ++p)
                x[p] = *reinterpret_cast<const bf8 *>(tile + plane_off8<ROWS, ROWB>(p, 16 * rb + m, 4 * c + q));
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const WFrag3 &wf = w[cb][C0 + c];
                lo[rb][cb] = mfma_bf16(wf.p[2], x[0], lo[rb][cb]);   // l h
                lo[rb][cb] = mfma_bf16(wf.p[0], x[2], lo[rb][cb]);   // h l
                lo[rb][cb] = mfma_bf16(wf.p[1], x[1], lo[rb][cb]);   // m m
                lo[rb][cb] = mfma_bf16(wf.p[1], x[0], lo[rb][cb]);   // m h
                lo[rb][cb] = mfma_bf16(wf.p[0], x[1], lo[rb][cb]);   // h m
                acc[rb][cb] = mfma_bf16(wf.p[0], x[0], acc[rb][cb]); // h h
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] += lo[rb][cb];
}
