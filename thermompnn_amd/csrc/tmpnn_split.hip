// Split-precision support kernels: the GEMM probe, the device self-test, the f16 weight-fragment images. The per-edge and node kernels
// live in tmpnn_edge.hip / tmpnn_msg.hip / tmpnn_edge_msg.hip / tmpnn_node.hip; the arithmetic is described in tmpnn_split.h.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "tmpnn_split.h"
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

// fragment image of one 128 x 128 block (see WImg in tmpnn_internal.h): [wv 8][c 4][plane 2][lane 64] x 16 B
// perm (full blocks only): the 8 values of lane group q in step c are k = 32 c + 4 q + {0..3} and 32 c + 16 + 4 q + {0..3} — the K order in
// which a wavefront that keeps its activations in the accumulator layout holds them (msg8_wave_kernel)
__global__ void prep_wimg_kernel(const float *__restrict__ W, int ld, int n_rows, int k_valid, int k_wrap, char *__restrict__ dst, bool perm) {
    const int idx = tm_bid() * tm_bdim() + tm_tid();      // (wv, c, lane)
    if (idx >= 8 * 4 * 64) return;
    const int lane = idx & 63, c = (idx >> 6) & 3, wv = idx >> 8, m = lane & 15, q = lane >> 4;
    const float *src = W + (size_t)(16 * wv + m) * ld + 32 * c + (perm ? 4 : 8) * q;
    // blocks with fewer than 128 rows / fewer than 128 columns (k_valid, a multiple of 8): zero fragments beyond, nothing read —
    // except the k_wrap columns after k_valid, which come from the same row one `ld` back (load_wfrag_split)
    const int k = 32 * c + 8 * q;
    const bool ok = 16 * wv + m < n_rows && k < k_valid + k_wrap;
    if (k >= k_valid) src -= ld;
    const f4 z = f4{0.f, 0.f, 0.f, 0.f};
    const f4 v0 = ok ? ld4(src) : z, v1 = ok ? ld4(src + (perm ? 16 : 4)) : z;
    unsigned w4[4][2];
    SplitH2::split2(f2{v0.x, v0.y}, w4[0]);
    SplitH2::split2(f2{v0.z, v0.w}, w4[1]);
    SplitH2::split2(f2{v1.x, v1.y}, w4[2]);
    SplitH2::split2(f2{v1.z, v1.w}, w4[3]);
#pragma unroll
    for (int p = 0; p < 2; ++p)
        *reinterpret_cast<u4 *>(dst + (size_t)wv * 8192 + c * 2048 + p * 1024 + lane * 16) = u4{w4[0][p], w4[1][p], w4[2][p], w4[3][p]};
}
int launch_prep_wimg(const float *W, int ld, char *dst, hipStream_t st, int n_rows, int k_valid, int k_wrap, bool perm) {
    prep_wimg_kernel<<<8, 256, 0, st>>>(W, ld, n_rows, k_valid, k_wrap, dst, perm);
    return tm_check_launch("prep_wimg");
}
