// Does VALU work issued between MFMAs overlap with them (one wavefront per SIMD)?  f32 16x16x4 vs bf16 16x16x32.
// hipcc --offload-arch=gfx950 -O3 tools/probe/overlap_probe.hip -o tools/probe/overlap_probe && ./overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV, int NT = 256>   // MODE 0: f32 mfma, 1: bf16 mfma, 2: no mfma.  NV = independent v_fma per MFMA slot
__global__ __launch_bounds__(NT, 1) void probe(int iters, unsigned long long *out, float *sink) {
    f4 acc[6];
    for (int k = 0; k < 6; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
    float a = 1e-3f * threadIdx.x, b = 1.0f + 1e-6f * threadIdx.x;
    bf8 ab, bb;
    for (int k = 0; k < 8; ++k) { ab[k] = (__bf16)(1e-3f * (threadIdx.x + k)); bb[k] = (__bf16)(1.0f + 1e-3f * k); }
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = 1e-3f * (threadIdx.x + k);
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (MODE == 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
                if (MODE == 1) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[k], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; ++j) v[(k + j) & 7] = __builtin_fmaf(v[(k + j) & 7], 1.0001f, 1e-7f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < 6; ++k) s += acc[k].x + acc[k].w;
    for (int k = 0; k < 8; ++k) s += v[k];
    if (s == 123.456f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
}

template <int MODE, int NV, int NT = 256>
void run(const char *name) {
    unsigned long long *out; float *sink;
    hipMalloc(&out, 256 * 8); hipMalloc(&sink, 1024);
    const int iters = 4000;
    probe<MODE, NV, NT><<<256, NT>>>(iters, out, sink);
    probe<MODE, NV, NT><<<256, NT>>>(iters, out, sink);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < 256; ++i) cyc += h[i];
    cyc /= 256;
    printf("%-10s NV=%d waves/SIMD=%d: %.1f cycles per slot per wave\n", name, NV, NT / 256, cyc / (iters * 24.0));
    hipFree(out); hipFree(sink);
}
int main() {
    run<2, 2>("valu only"); run<2, 4>("valu only"); run<2, 8>("valu only");
    run<0, 0>("f32 mfma"); run<0, 2>("f32 mfma"); run<0, 4>("f32 mfma"); run<0, 8>("f32 mfma");
    run<1, 0>("bf16 mfma"); run<1, 2>("bf16 mfma"); run<1, 4>("bf16 mfma"); run<1, 8>("bf16 mfma");
    printf("---- two wavefronts per SIMD, each running the same stream ----\n");
    run<2, 4, 512>("valu only"); run<2, 8, 512>("valu only");
    run<0, 0, 512>("f32 mfma"); run<0, 4, 512>("f32 mfma"); run<0, 8, 512>("f32 mfma");
    run<1, 0, 512>("bf16 mfma"); run<1, 4, 512>("bf16 mfma"); run<1, 8, 512>("bf16 mfma");
    return 0;
}
