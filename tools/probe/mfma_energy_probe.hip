// What does an instruction class cost at the board's power cap? (round 5, docs/NOTEBOOK.md 9.13)
// Every CU runs 8 wavefronts (2 per SIMD) of ONE instruction class back to back on register operands for ~1 s; per class: the rate by HIP
// events and the shader clock the chip kept (cycle counter against the 100 MHz reference, wavefront 0 of workgroup 0). tools/power_probe
// style rocm-smi samples beside it give the package power. At the cap, the clock a stream can hold IS its energy per instruction.
//    hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_energy_probe tools/probe/mfma_energy_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// 0: v_mfma_f32_16x16x32_f16   1: v_mfma_f32_32x32x16_f16   2: v_pk_fma_f32   3: v_fma_f32   4: v_exp_f32   5: ds_read_b128 (conflict-free)
// 6: v_cvt_pk_f16_f32 + v_cvt_f32_f16 (the split's conversions)
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(int iters, float *sink, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    if (SHAPE == 5) { for (int o = threadIdx.x * 16; o < 65536; o += 512 * 16) *reinterpret_cast<u4 *>(lds + o) = u4{1u, 2u, 3u, (unsigned)o}; __syncthreads(); }
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    float out = 0.f;
    if (SHAPE == 0) {
        f4 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
        }
        out = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
    } else if (SHAPE == 1) {
        f16v acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
        }
        out = acc[0][0] + acc[1][5] + acc[2][10] + acc[3][15];
    } else if (SHAPE == 2) {
        f2 x[8];
        for (int u = 0; u < 8; ++u) x[u] = f2{0.5f + 0.01f * lane, 0.25f + u};
        const f2 m = f2{0.999f, 1.001f}, c = f2{1e-3f, -1e-3f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(m), "v"(c));
        }
        for (int u = 0; u < 8; ++u) out += x[u].x + x[u].y;
    } else if (SHAPE == 3) {
        float x[8];
        for (int u = 0; u < 8; ++u) x[u] = 0.5f + 0.01f * lane + u;
        const float m = 0.999f, c = 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(m), "v"(c));
        }
        for (int u = 0; u < 8; ++u) out += x[u];
    } else if (SHAPE == 4) {
        float x[8];
        for (int u = 0; u < 8; ++u) x[u] = -0.5f - 0.01f * lane - u;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("v_exp_f32 %0, %0\n\ts_nop 0" : "+v"(x[u]));
        }
        for (int u = 0; u < 8; ++u) out += x[u];
    } else if (SHAPE == 5) {
        u4 x[8];
        const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) void *)lds + 16u * lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[u]) : "v"(base), "n"(1024 * u));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        for (int u = 0; u < 8; ++u) out += (float)(x[u].x + x[u].w);
    } else {
        float x[8];
        for (int u = 0; u < 8; ++u) x[u] = 0.5f + 0.01f * lane + u;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    unsigned p;
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(x[u]), "v"(x[u + 1]));
                    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[u]) : "v"(p));
                    asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x[u + 1]) : "v"(p));
                }
        }
        for (int u = 0; u < 8; ++u) out += x[u];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    if (out == 123.456f) sink[threadIdx.x] = out;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

int main(int argc, char **argv) {
    float *sink; unsigned long long *clk, h[2];
    (void)hipMalloc(&sink, 4096); (void)hipMalloc(&clk, 16);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const char *name[7] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_pk_fma_f32", "v_fma_f32", "v_exp_f32", "ds_read_b128", "cvt_pk_f16 + 2 cvt_f32_f16"};
    const int per_iter[7] = {8, 4, 32, 32, 16, 8, 12};
    const int iters_of[7] = {8000000, 8000000, 6000000, 8000000, 7000000, 10000000, 9000000};     // ~1 s per launch: the power manager settles
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int shape = 0; shape < 7; ++shape) {
        const int iters = iters_of[shape];
        (void)hipEventRecord(e0);
        switch (shape) {
            case 0: k<0><<<cus, 512>>>(iters, sink, clk); break;
            case 1: k<1><<<cus, 512>>>(iters, sink, clk); break;
            case 2: k<2><<<cus, 512>>>(iters, sink, clk); break;
            case 3: k<3><<<cus, 512>>>(iters, sink, clk); break;
            case 4: k<4><<<cus, 512>>>(iters, sink, clk); break;
            case 5: k<5><<<cus, 512>>>(iters, sink, clk); break;
            default: k<6><<<cus, 512>>>(iters, sink, clk); break;
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double n = (double)cus * 8 * iters * per_iter[shape];          // wavefront-instructions of the launch
        printf("%-28s %7.1f ms  %8.2f G wavefront-instr/s  clock %.3f GHz  %6.2f SIMD cycles per instruction (launch) %6.2f (wavefront 0 alone)",
               name[shape], ms, n / (ms * 1e-3) / 1e9, h[0] / (h[1] * 10.0), (ms * 1e-3) * (h[0] / (h[1] * 10.0)) * 1e9 / (n / (cus * 4.0)),
               (double)h[0] / ((double)iters * per_iter[shape]));
        if (shape < 2) printf("  %.0f TFLOP/s", n * (shape == 0 ? 16384.0 : 32768.0) / (ms * 1e-3) / 1e12);
        printf("\n");
        fflush(stdout);
        if (argc > 1) { (void)hipDeviceSynchronize(); struct timespec ts = {0, 300000000}; nanosleep(&ts, nullptr); }
    }
    return 0;
}
