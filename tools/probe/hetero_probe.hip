// Do MFMAs of one wavefront overlap with VALU work of ANOTHER wavefront on the same SIMD (f16 16x16x32, 4 passes)?
// 8 wavefronts per workgroup (2 per SIMD): wavefronts 0-3 run a pure MFMA stream, wavefronts 4-7 a pure VALU stream
// (v_fma_f32 or packed v_pk_fma_f32). Each role is timed alone and together.
// hipcc --offload-arch=gfx950 -O3 tools/probe/hetero_probe.hip -o tools/probe/hetero_probe && tools/probe/hetero_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int PK>
__global__ __launch_bounds__(512, 1) void hetero(int iters, int run_mfma, int run_valu, unsigned long long *out, float *sink) {
    const int wv = threadIdx.x >> 6;
    unsigned long long c0 = 0, c1 = 0;
    float s = 0.f;
    if (wv < 4) {
        if (run_mfma) {
            f4 acc[6];
            for (int k = 0; k < 6; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
            h8 a, b;
            for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(1e-3f * (threadIdx.x + k)); b[k] = (_Float16)(1.0f + 1e-3f * k); }
            c0 = __builtin_readcyclecounter();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int k = 0; k < 6; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
            }
            c1 = __builtin_readcyclecounter();
            for (int k = 0; k < 6; ++k) s += acc[k].x + acc[k].w;
        }
    } else if (run_valu) {
        f2 v[8];
        for (int k = 0; k < 8; ++k) v[k] = f2{1e-3f * (threadIdx.x + k), 2e-3f * (threadIdx.x + k)};
        c0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 12; ++g)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (PK) v[k] = __builtin_elementwise_fma(v[k], f2{1.0001f, 1.0001f}, f2{1e-7f, 1e-7f});
                    else v[k].x = __builtin_fmaf(v[k].x, 1.0001f, 1e-7f);
                }
        }
        c1 = __builtin_readcyclecounter();
        for (int k = 0; k < 8; ++k) s += v[k].x + v[k].y;
    }
    if (s == 123.456f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wv] = c1 - c0;
}

template <int PK>
void run(const char *name, int m, int v) {
    unsigned long long *out; float *sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 4096);
    const int iters = 2000;
    hetero<PK><<<256, 512>>>(iters, m, v, out, sink);
    hetero<PK><<<256, 512>>>(iters, m, v, out, sink);
    hipDeviceSynchronize();
    static unsigned long long h[2048];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double cm = 0, cv = 0;
    for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) cm += h[b * 8 + w]; for (int w = 4; w < 8; ++w) cv += h[b * 8 + w]; }
    cm /= 1024; cv /= 1024;
    printf("%-28s mfma wavefront: %6.1f cycles / MFMA     valu wavefront: %6.2f cycles / VALU op\n", name, cm / (iters * 24.0), cv / (iters * 96.0));
    hipFree(out); hipFree(sink);
}
int main() {
    run<0>("mfma alone", 1, 0);
    run<0>("v_fma_f32 alone", 0, 1);
    run<0>("mfma + v_fma_f32", 1, 1);
    run<1>("v_pk_fma_f32 alone", 0, 1);
    run<1>("mfma + v_pk_fma_f32", 1, 1);
    return 0;
}
