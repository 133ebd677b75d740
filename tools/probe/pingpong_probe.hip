// Does phase opposition of the two wavefronts of a SIMD pay for the per-edge kernels' real building blocks?
// One workgroup of 8 wavefronts per CU runs R x { G: mma_tile_split (48 x 128 x 128, f16x2, 36 MFMAs per wavefront);
// barrier; V: 3 x gelu4 + 3 x store_split (the GELU + split epilogue); barrier }.
//   MODE 0  all 8 wavefronts in the same phase (what the shipped kernels do)
//   MODE 1  wavefronts 4-7 run the same program one barrier late: on every SIMD one wavefront is in G while the other is in V
//   MODE 2/3 = G only / V only (the parts)
// hipcc --offload-arch=gfx950 -O3 -mno-amdgpu-ieee -fno-honor-nans -I thermompnn_amd/csrc tools/probe/pingpong_probe.hip -o tools/probe/pingpong_probe
#include <stdio.h>
#include "tmpnn_split.h"

// G with the three row blocks' accumulators interleaved term by term: no MFMA depends on the one issued just before it
__device__ __forceinline__ void mma_ilv(const char *tile, const WFragS<SplitH2> (&w)[1][4], f4 (&acc)[3][1], int lane) {
    const int m = lane & 15, q = lane >> 4;
    u4 x[2][3][2];
#define RD(c, b)                                                                                                   \
    for (int rb = 0; rb < 3; ++rb)                                                                                 \
        for (int p = 0; p < 2; ++p) x[b][rb][p] = *reinterpret_cast<const u4 *>(tile + plane_off8<>(p, 16 * rb + m, 4 * (c) + q));
#define TM_HF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0)
#pragma unroll
    RD(0, 0)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c + 1 < 4) {
#pragma unroll
            RD(c + 1, (c + 1) & 1)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = TM_HF(w[0][c].p[1], x[c & 1][rb][0], acc[rb][0]);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = TM_HF(w[0][c].p[0], x[c & 1][rb][1], acc[rb][0]);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = TM_HF(w[0][c].p[0], x[c & 1][rb][0], acc[rb][0]);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef RD
#undef TM_HF
}

template <int MODE, int PF>
__global__ __launch_bounds__(512, 2) void pp_kernel(const float *__restrict__ W, float *__restrict__ Y, int reps, unsigned long long *cyc) {
    using SP = SplitH2;
    __shared__ __attribute__((aligned(16))) char tA[2][2 * SPLIT_PLANE_BYTES];
    __shared__ __attribute__((aligned(16))) char tB[2][2 * SPLIT_PLANE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int c4 = 4 * wv + q;
    WFragS<SP> w[1][4];
    load_wfrag_split<SP, 4>(W, TM_H, 16 * wv, 0, TM_H, w[0], lane);
    for (int t = 0; t < 2; ++t)
        for (int rb = 0; rb < 3; ++rb) {
            const f4 v = f4{0.01f * (lane + rb), -0.02f * (wv + t), 0.003f * m, 0.5f - 0.01f * q};
            store_split<SP>(tA[t], 16 * rb + m, c4, v);
            store_split<SP>(tB[t], 16 * rb + m, c4, v);
        }
    __syncthreads();
    const bool late = MODE == 1 && wv >= 4;
    const unsigned long long c0 = __builtin_readcyclecounter();
    if (late) __syncthreads();
    f4 keep = f4{0.f, 0.f, 0.f, 0.f};
    unsigned long long ph[4] = {0, 0, 0, 0}, tl = __builtin_readcyclecounter();
    auto mark = [&](int k) { const unsigned long long t = __builtin_readcyclecounter(); ph[k] += t - tl; tl = t; };
    for (int r = 0; r < reps; ++r) {
        const int t = r & 1;
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) { acc[rb][0] = f4{0.1f, 0.2f, 0.3f, 0.4f} + keep; touch(acc[rb][0]); }
        if (MODE != 3) { if (PF == 99) mma_ilv(tA[t], w, acc, lane); else mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, PF == 99 ? 0 : PF>(tA[t], w, acc, lane); }
        __builtin_amdgcn_sched_barrier(0);
        mark(0);
        if (MODE < 2) __syncthreads();
        mark(1);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 2) {
            f4 g[3];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0] + keep);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) store_split<SP>(tB[t], 16 * rb + m, c4, g[rb]);
            keep = g[0] * 1e-3f;
        } else keep += acc[0][0] + acc[1][0] + acc[2][0];
        __builtin_amdgcn_sched_barrier(0);
        mark(2);
        if (MODE < 2) __syncthreads();
        mark(3);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1 && !late) __syncthreads();
    const unsigned long long c1 = __builtin_readcyclecounter();
    st4(Y + ((size_t)blockIdx.x * 512 + tid) * 4, keep);
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
    if (blockIdx.x == 7 && lane == 0 && (wv == 0 || wv == 4)) for (int k = 0; k < 4; ++k) cyc[256 + (wv >> 2) * 4 + k] = ph[k];
}

template <int MODE, int PF>
double run(const float *W, float *Y, unsigned long long *cyc, int reps) {
    pp_kernel<MODE, PF><<<256, 512>>>(W, Y, reps, cyc);
    pp_kernel<MODE, PF><<<256, 512>>>(W, Y, reps, cyc);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += h[i];
    unsigned long long ph[8];
    hipMemcpy(ph, cyc + 256, sizeof(ph), hipMemcpyDeviceToHost);
    printf("   mode %d pf %d: wave0 G %.0f wait %.0f V %.0f wait %.0f | wave4 G %.0f wait %.0f V %.0f wait %.0f\n", MODE, PF, ph[0] / (double)reps, ph[1] / (double)reps,
           ph[2] / (double)reps, ph[3] / (double)reps, ph[4] / (double)reps, ph[5] / (double)reps, ph[6] / (double)reps, ph[7] / (double)reps);
    return s / 256 / reps;
}

int main() {
    float *W, *Y; unsigned long long *cyc;
    hipMalloc(&W, 128 * 128 * 4); hipMalloc(&Y, 256 * 512 * 16); hipMalloc(&cyc, 264 * 8);
    float hw[128 * 128];
    for (int i = 0; i < 128 * 128; ++i) hw[i] = 0.05f * ((i * 37 % 101) - 50) / 50.f;
    hipMemcpy(W, hw, sizeof(hw), hipMemcpyHostToDevice);
    const int reps = 2000;
    {
        const double g = run<2, 0>(W, Y, cyc, reps), v = run<3, 0>(W, Y, cyc, reps), s = run<0, 0>(W, Y, cyc, reps), p = run<1, 0>(W, Y, cyc, reps);
        printf("PF 0: cycles per (G + V) round of one workgroup: G only %.0f, V only %.0f, same phase %.0f, opposed %.0f (%.1f %% faster)\n",
               g, v, s, p, 100.0 * (s - p) / s);
    }
    {
        const double g = run<2, 3>(W, Y, cyc, reps), v = run<3, 3>(W, Y, cyc, reps), s = run<0, 3>(W, Y, cyc, reps), p = run<1, 3>(W, Y, cyc, reps);
        printf("PF 3: cycles per (G + V) round of one workgroup: G only %.0f, V only %.0f, same phase %.0f, opposed %.0f (%.1f %% faster)\n",
               g, v, s, p, 100.0 * (s - p) / s);
    }
    {
        const double g = run<2, 99>(W, Y, cyc, reps), v = run<3, 99>(W, Y, cyc, reps), s = run<0, 99>(W, Y, cyc, reps), p = run<1, 99>(W, Y, cyc, reps);
        printf("ILV : cycles per (G + V) round of one workgroup: G only %.0f, V only %.0f, same phase %.0f, opposed %.0f (%.1f %% faster)\n",
               g, v, s, p, 100.0 * (s - p) / s);
    }
    return 0;
}
