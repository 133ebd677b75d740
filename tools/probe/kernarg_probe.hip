// Does kernel-argument preloading (gfx950: the CP writes the first kernarg dwords into user SGPRs before the wavefront
// starts; -mllvm -amdgpu-kernarg-preload-count=N) shorten a small launch?  Three dependent round trips start a kernel of
// this library (kernarg -> pointer -> first row); this probe chains launches of a kernel with that shape, once with the
// arguments in a by-value struct (never preloaded) and once flat (preloaded when built with the flag).
//   hipcc --offload-arch=gfx950 -O3 -mcode-object-version=5 -mllvm -amdgpu-kernarg-preload-count=16 -o kernarg_probe kernarg_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { const int* idx; const float* src; float* dst; int n; int pad; };
__global__ __launch_bounds__(512) void k_struct(Args a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) a.dst[i] = a.src[a.idx[i]] + 1.0f;
}
__global__ __launch_bounds__(512) void k_flat(const int* idx, const float* src, float* dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]] + 1.0f;
}
int main() {
    const int n = 256 * 512;
    int* idx; float *a, *b;
    hipMalloc(&idx, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemset(idx, 0, n * 4); hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int form = 0; form < 2; ++form)
        for (int rep = 0; rep < 3; ++rep) {
            const int chains = 300, per = 20;
            auto run = [&]() {
                for (int i = 0; i < per; ++i) {
                    float* src = (i & 1) ? b : a; float* dst = (i & 1) ? a : b;
                    if (form == 0) { Args g{idx, src, dst, n, 0}; hipLaunchKernelGGL(k_struct, dim3(256), dim3(512), 0, s, g); }
                    else hipLaunchKernelGGL(k_flat, dim3(256), dim3(512), 0, s, idx, src, dst, n);
                }
            };
            for (int c = 0; c < 20; ++c) run();
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int c = 0; c < chains; ++c) run();
            hipStreamSynchronize(s);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%s  %.3f us per launch (chains of %d dependent launches)\n", form ? "flat  " : "struct", us / chains / per, per);
        }
    return 0;
}
