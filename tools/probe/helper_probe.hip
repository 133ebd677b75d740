// Helper wavefronts (round 5): does a THIRD wavefront per SIMD do the per-edge kernels' non-GEMM work (incoming-tile split, LayerNorm rows,
// stores: LDS -> VALU -> LDS / global) in the shadow of the compute wavefronts' MFMA phases?
// Round of the compute wavefronts (0-7, 16 output columns each): tile GEMM 48 x 128 x 128 (f16x2) -> GELU + split -> barrier, as ilv_probe
// FORM 0. EXTRA work unit = ds_read_b128 of an fp32 row chunk -> gelu4 -> store_split into a plane tile (36 VALU + 1 read + 1 write:
// the shape of split_tile / the LayerNorm row phase).
//   MODE 0  8 wavefronts, no extra work                                   (512 threads)
//   MODE 2  8 wavefronts, each also does E extra units per round          (512 threads: what the shipped kernels do)
//   MODE 1  12 wavefronts: 0-7 compute only, 8-11 do 2 E extra units each (768 threads, <= 168 VGPRs: three wavefronts per SIMD)
//   MODE 3  12 wavefronts, helpers idle at the barrier only               (the cost of their presence)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -DTM_GELU_NAN3=1 -I thermompnn_amd/csrc tools/probe/helper_probe.hip -o tools/probe/helper_probe
#include <stdio.h>

#include "tmpnn_split.h"

template <int MODE, int E>
__global__ __launch_bounds__(MODE == 1 || MODE == 3 ? 768 : 512) void helper_kernel(const float *__restrict__ W, float *__restrict__ Y, int reps, unsigned long long *cyc) {
    using SP = SplitH2;
    __shared__ __attribute__((aligned(16))) char tA[2][2 * SPLIT_PLANE_BYTES];
    __shared__ __attribute__((aligned(16))) float tF[TM_TILE * TM_H];          // fp32 rows the extra units read
    __shared__ __attribute__((aligned(16))) char tX[2 * SPLIT_PLANE_BYTES];    // planes the extra units write
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, q = lane >> 4;
    const int c4 = 4 * (wv & 7) + q;
    WFragS<SP> w[1][4];
    if (wv < 8) load_wfrag_split<SP, 4>(W, TM_H, 16 * wv, 0, TM_H, w[0], lane);
    for (int idx = tid; idx < TM_TILE * 32; idx += blockDim.x) {
        const int row = idx >> 5, c = idx & 31;
        const f4 v = f4{0.01f * (c + row), -0.02f * row, 0.003f * c, 0.5f - 0.01f * (row & 7)};
        store_split<SP>(tA[0], row, c, v);
        store_split<SP>(tA[1], row, c, v);
        st4(tF + chunk_off(row, c), v);
    }
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    f4 keep = f4{0.f, 0.f, 0.f, 0.f};
    auto extra = [&](int unit) {            // unit < 48 * 32 / 64 = 24 distinct 64-chunk groups
        const int idx = (unit * 64 + lane) % (TM_TILE * 32), row = idx >> 5, c = idx & 31;
        const f4 g = gelu4(ld4(tF + chunk_off(row, c)) + keep * 1e-3f);
        store_split<SP>(tX, row, c, g);
        keep += g * 1e-3f;
    };
    for (int r = 0; r < reps; ++r) {
        const int t = r & 1;
        if (wv < 8) {
            f4 init = f4{0.1f, 0.2f, 0.3f, 0.4f} + keep * 1e-3f;
            touch(init);
            f4 acc[3][1];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = init * (1.0f + 0.25f * rb);
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, 2>(tA[t], w, acc, lane);
            __builtin_amdgcn_sched_barrier(0);
            f4 g[3];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) store_split<SP>(tA[t ^ 1], 16 * rb + m, c4, g[rb]);
            keep = g[2];
            if (MODE == 2) {
#pragma unroll
                for (int u = 0; u < E; ++u) extra(wv * E + u);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 2 * E; ++u) extra((wv - 8) * 2 * E + u);
        }
        __syncthreads();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    st4(Y + ((size_t)blockIdx.x * 768 + tid) * 4, keep);
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int MODE, int E>
double run(const float *W, float *Y, unsigned long long *cyc, int reps) {
    const int nt = MODE == 1 || MODE == 3 ? 768 : 512;
    helper_kernel<MODE, E><<<256, nt>>>(W, Y, reps, cyc);
    helper_kernel<MODE, E><<<256, nt>>>(W, Y, reps, cyc);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += h[i];
    return s / 256 / reps;
}

int main() {
    float *W, *Y; unsigned long long *cyc;
    (void)hipMalloc(&W, 128 * 128 * 4); (void)hipMalloc(&Y, 256 * 768 * 16); (void)hipMalloc(&cyc, 264 * 8);
    static float hw[128 * 128];
    for (int i = 0; i < 128 * 128; ++i) hw[i] = 0.05f * ((i * 37 % 101) - 50) / 50.f;
    (void)hipMemcpy(W, hw, sizeof(hw), hipMemcpyHostToDevice);
    const int reps = 2000;
    const double base = run<0, 1>(W, Y, cyc, reps), idle = run<3, 1>(W, Y, cyc, reps);
    printf("8 wavefronts, no extra work: %.0f cycles per round; 12 wavefronts, helpers only at the barrier: %.0f\n", base, idle);
#define ROW(E) { const double own = run<2, E>(W, Y, cyc, reps), hlp = run<1, E>(W, Y, cyc, reps); \
    printf("extra work = %d unit(s) of 36 VALU per compute wavefront and round: done by the compute wavefronts %.0f (+%.0f), by four helper wavefronts %.0f (+%.0f)\n", E, own, own - base, hlp, hlp - base); }
    ROW(1) ROW(2) ROW(3) ROW(4)
    return 0;
}
