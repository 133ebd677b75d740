// Wavefront SPECIALISATION for the per-edge kernels' GEMM -> GELU/split chain (round 5): do a pure-MFMA wavefront and a pure-VALU
// wavefront on the same SIMD overlap on the real building blocks, as tools/probe/hetero_probe.hip says two synthetic streams do?
// One workgroup of 8 wavefronts per CU (wavefronts w and w + 4 share a SIMD), one barrier per round, two tile chains in flight:
//   wavefronts 0-3 ("M"): round r = tile GEMM 48 x 128 x 128 (f16x2) from plane tile P[r & 1], 32 output columns each (two
//                         16-column blocks sharing every B fragment: half the LDS fragment reads per MAC), fp32 result -> F[r & 1]
//   wavefronts 4-7 ("V"): round r = GELU + split of F[(r - 1) & 1] -> plane tile P[(r + 1) & 1]  (what the M wavefronts read next round)
// so the GEMM of round r + 1 consumes the epilogue of the GEMM of round r - 1: two interleaved dependency chains = two residues in
// flight per workgroup. Work per round = the shipped kernels' "GEMM + epilogue + barrier" round (ilv_probe FORM 0).
//   MODE 0 both roles   1 M only (V wavefronts only keep the barriers)   2 V only
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -DTM_GELU_NAN3=1 -I thermompnn_amd/csrc tools/probe/spec_probe.hip -o tools/probe/spec_probe
#include <stdio.h>

#include "tmpnn_split.h"

#ifndef SPEC_PF
#define SPEC_PF 2
#endif

template <int MODE>
__global__ __launch_bounds__(512, 2) void spec_kernel(const float *__restrict__ W, float *__restrict__ Y, int reps, unsigned long long *cyc) {
    using SP = SplitH2;
    __shared__ __attribute__((aligned(16))) char tP[2][2 * SPLIT_PLANE_BYTES];
    __shared__ __attribute__((aligned(16))) float tF[2][TM_TILE * TM_H];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    f4 keep = f4{0.f, 0.f, 0.f, 0.f};
    // common set-up: both plane tiles and both fp32 tiles hold something finite
    for (int t = 0; t < 2; ++t)
        for (int it = 0; it < 3; ++it) {
            const int idx = it * 512 + tid, row = idx >> 5, c = idx & 31;
            const f4 v = f4{0.01f * (lane + it), -0.02f * (wv + t), 0.003f * m, 0.5f - 0.01f * q};
            store_split<SP>(tP[t], row, c, v);
            st4(tF[t] + chunk_off(row, c), v);
        }
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    if (wv < 4) {
        WFragS<SP> w[2][4];
        load_wfrag_split<SP, 4>(W, TM_H, 32 * wv, 0, TM_H, w[0], lane);
        load_wfrag_split<SP, 4>(W, TM_H, 32 * wv + 16, 0, TM_H, w[1], lane);
        for (int r = 0; r < reps; ++r) {
            const int t = r & 1;
            if (MODE != 2) {
                f4 acc[3][2];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) { acc[rb][cb] = f4{0.1f, 0.2f, 0.3f, 0.4f} + keep * 1e-3f; }
                touch(acc[0][0]);
                mma_tile_split<SP, 4, 2, 3, TM_TILE, 256, 4, 0, true, SPEC_PF>(tP[t], w, acc, lane);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) st4(tF[t] + chunk_off(16 * rb + m, 8 * wv + 4 * cb + q), acc[rb][cb]);
                keep = acc[2][1];
            }
            __syncthreads();
        }
    } else {
        const int vt = tid - 256;
        for (int r = 0; r < reps; ++r) {
            const int t = r & 1;
            if (MODE != 1) {
                f4 v[6], g[6];
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int idx = it * 256 + vt;
                    v[it] = ld4(tF[t ^ 1] + chunk_off(idx >> 5, idx & 31));
                }
#pragma unroll
                for (int it = 0; it < 6; ++it) g[it] = gelu4(v[it] + keep * 1e-3f);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int idx = it * 256 + vt;
                    store_split<SP>(tP[t ^ 1], idx >> 5, idx & 31, g[it]);
                }
                keep = g[5];
            }
            __syncthreads();
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    st4(Y + ((size_t)blockIdx.x * 512 + tid) * 4, keep);
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int MODE>
double run(const float *W, float *Y, unsigned long long *cyc, int reps) {
    spec_kernel<MODE><<<256, 512>>>(W, Y, reps, cyc);
    spec_kernel<MODE><<<256, 512>>>(W, Y, reps, cyc);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += h[i];
    return s / 256 / reps;
}

int main() {
    float *W, *Y; unsigned long long *cyc;
    (void)hipMalloc(&W, 128 * 128 * 4); (void)hipMalloc(&Y, 256 * 512 * 16); (void)hipMalloc(&cyc, 264 * 8);
    static float hw[128 * 128];
    for (int i = 0; i < 128 * 128; ++i) hw[i] = 0.05f * ((i * 37 % 101) - 50) / 50.f;
    (void)hipMemcpy(W, hw, sizeof(hw), hipMemcpyHostToDevice);
    const int reps = 2000;
    const double both = run<0>(W, Y, cyc, reps), mo = run<1>(W, Y, cyc, reps), vo = run<2>(W, Y, cyc, reps);
    printf("specialised wavefronts, PF %d: cycles per round: both roles %.0f, M wavefronts only %.0f, V wavefronts only %.0f\n", SPEC_PF, both, mo, vo);
    return 0;
}
