// Why does "epilogue + barrier" take twice the epilogue (ilv_probe FORM 6 vs 8: 1489 vs 697 cycles per round)? Hypothesis: after a
// barrier the two wavefronts of a SIMD start the same instruction stream in the same cycle, the older one wins the VALU arbitration
// and the younger one crawls; the older then idles at the next barrier while the younger finishes alone at the single-wavefront
// issue rate (one VALU instruction per ~5.4 cycles). This probe stamps when wavefronts 0 and 4 (one SIMD) reach the barrier and
// tries the knobs the ISA offers: s_setprio (static / alternating), a start skew.
//   GEMM = 0: round = GELU + split epilogue + barrier;  1: round = tile GEMM + epilogue + barrier (the shipped round)
//   PRIO  0 none | 1 wavefronts 4-7 at s_setprio 1 | 2 wavefronts 0-3 at s_setprio 1 | 3 alternating by round | 4 wavefronts 4-7 start
//         the round ~64 cycles late (s_sleep 1) | 5 wavefronts 4-7 at prio 1 for the first half of the epilogue only | 6 wavefronts 4-7
//         at s_setprio 3
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -DTM_GELU_NAN3=1 -I thermompnn_amd/csrc tools/probe/prio_probe.hip -o tools/probe/prio_probe
#include <stdio.h>

#include "tmpnn_split.h"

// GELU with SCALAR fmas (same arithmetic as gelu2, one value per instruction): v_pk_fma_f32 streams of two wavefronts share a SIMD
// unfairly (625 vs 1115 cycles per 128 instructions, stream_kernel below), v_fma_f32 streams fairly (641 / 648)
__device__ __forceinline__ float sfma(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ float gelu1s(float x) {
    const float t = __builtin_elementwise_minimum(fabsf(x), 5.656854249f);
    float q = __builtin_fmaf(3.309543916e-05f, t, -7.692427171e-04f);
    q = sfma(q, t, 8.080792133e-03f);
    q = sfma(q, t, -5.341222090e-02f);
    q = sfma(q, t, -4.587708865e-01f);
    q = sfma(q, t, -1.151201730e+00f);
    const float e = sfma(q, t, -9.999930581e-01f);
    return __builtin_fmaf(-t, __builtin_amdgcn_exp2f(e), __builtin_elementwise_maximum(x, 0.f));
}
__device__ __forceinline__ f4 gelu4s(f4 v) { return f4{gelu1s(v.x), gelu1s(v.y), gelu1s(v.z), gelu1s(v.w)}; }

template <int GEMM, int PRIO>
__global__ __launch_bounds__(512, 2) void prio_kernel(const float *__restrict__ W, float *__restrict__ Y, int reps, unsigned long long *cyc) {
    using SP = SplitH2;
    __shared__ __attribute__((aligned(16))) char tA[2][2 * SPLIT_PLANE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int c4 = 4 * wv + q;
    const bool young = __builtin_amdgcn_readfirstlane(tid) >= 256;
    WFragS<SP> w[1][4];
    load_wfrag_split<SP, 4>(W, TM_H, 16 * wv, 0, TM_H, w[0], lane);
    for (int t = 0; t < 2; ++t)
        for (int rb = 0; rb < 3; ++rb) {
            const f4 v = f4{0.01f * (lane + rb), -0.02f * (wv + t), 0.003f * m, 0.5f - 0.01f * q};
            store_split<SP>(tA[t], 16 * rb + m, c4, v);
        }
    if (PRIO == 1 && young) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 6 && young) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2 && !young) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    f4 keep = f4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t_arrive = 0, t_start = c0;
    for (int r = 0; r < reps; ++r) {
        const int t = r & 1;
        if (PRIO == 3) { if (young == (bool)(r & 1)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        if (PRIO == 4 && young) __builtin_amdgcn_s_sleep(1);
        if (PRIO == 7 && young) __builtin_amdgcn_s_sleep(3);
        if (PRIO == 8 && young) __builtin_amdgcn_s_sleep(6);
        if (PRIO == 5 && young) __builtin_amdgcn_s_setprio(1);
        f4 init = f4{0.1f, 0.2f, 0.3f, 0.4f} + keep * 1e-3f;
        touch(init);
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = init * (1.0f + 0.25f * rb);
        if (GEMM) mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, 2>(tA[t], w, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        f4 g[3];
        if (PRIO == 9 && young) {       // the younger wavefronts run the epilogue one row block at a time: another instruction-class sequence
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                g[rb] = gelu4(acc[rb][0]);
                store_split<SP>(tA[t ^ 1], 16 * rb + m, c4, g[rb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) g[rb] = PRIO == 10 ? gelu4s(acc[rb][0]) : gelu4(acc[rb][0]);
            __builtin_amdgcn_sched_barrier(0);
            if (PRIO == 5 && young) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) store_split<SP>(tA[t ^ 1], 16 * rb + m, c4, g[rb]);
        }
        keep = g[2];
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ta = __builtin_readcyclecounter();
        t_arrive += ta - t_start;
        __syncthreads();
        t_start = __builtin_readcyclecounter();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    st4(Y + ((size_t)blockIdx.x * 512 + tid) * 4, keep);
    if (lane == 0) {
        cyc[blockIdx.x * 16 + wv] = t_arrive;
        if (wv == 0) cyc[blockIdx.x * 16 + 8] = c1 - c0;
    }
}

template <int GEMM, int PRIO>
void run(const float *W, float *Y, unsigned long long *cyc, int reps, const char *name) {
    prio_kernel<GEMM, PRIO><<<256, 512>>>(W, Y, reps, cyc);
    prio_kernel<GEMM, PRIO><<<256, 512>>>(W, Y, reps, cyc);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 16];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double tot = 0, arr[8] = {0};
    for (int b = 0; b < 256; ++b) {
        tot += h[b * 16 + 8];
        for (int w = 0; w < 8; ++w) arr[w] += h[b * 16 + w];
    }
    printf("%s %-34s round %6.0f cycles | barrier reached after (wavefront 0..7): ", GEMM ? "GEMM + epilogue" : "epilogue only  ", name, tot / 256 / reps);
    for (int w = 0; w < 8; ++w) printf("%5.0f ", arr[w] / 256 / reps);
    printf("\n");
}

// Homogeneous streams, with and without a barrier every 128 instructions: is the older / younger asymmetry a property of the VALU
// arbitration itself (any stream), or of something in the epilogue (transcendentals, LDS writes, dependent chains)?
//   CLS 0: v_pk_fma_f32 on 8 independent registers   1: v_fma_f32   2: 7 x v_pk_fma_f32 + 1 x v_exp_f32   3: pk_fma as ONE dependent chain pair
template <int CLS, int BAR>
__global__ __launch_bounds__(512, 2) void stream_kernel(int iters, unsigned long long *cyc, float *sink) {
    const int wv = threadIdx.x >> 6;
    f2 p[8];
    float a[8];
    const float x = 0.9999f + 1e-7f * threadIdx.x, y = 1e-7f * (threadIdx.x + 1);
    const f2 px = f2{x, x}, py = f2{y, y};
    for (int k = 0; k < 8; ++k) { a[k] = 1e-3f * (threadIdx.x + k); p[k] = f2{a[k], 0.5f + 1e-3f * k}; }
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long busy = 0, ts = c0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (CLS == 0) asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                                       "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9"
                                       : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(px), "v"(py));
            if (CLS == 1) asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                                       "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                                       : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y));
            if (CLS == 2) asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                                       "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_exp_f32 %10, %10"
                                       : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(px), "v"(py), "v"(a[r & 7]));
            if (CLS == 3) asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\t"
                                       "v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9"
                                       : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(px), "v"(py));
        }
        if (BAR) {
            const unsigned long long ta = __builtin_readcyclecounter();
            busy += ta - ts;
            __syncthreads();
            ts = __builtin_readcyclecounter();
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y;
    if (s == 123.456f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 16 + wv] = BAR ? busy : c1 - c0; if (wv == 0) cyc[blockIdx.x * 16 + 8] = c1 - c0; }
}
template <int CLS, int BAR>
void run_stream(unsigned long long *cyc, float *sink, const char *name) {
    const int iters = 1000;
    stream_kernel<CLS, BAR><<<256, 512>>>(iters, cyc, sink);
    stream_kernel<CLS, BAR><<<256, 512>>>(iters, cyc, sink);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 16];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double tot = 0, arr[8] = {0};
    for (int b = 0; b < 256; ++b) { tot += h[b * 16 + 8]; for (int w = 0; w < 8; ++w) arr[w] += h[b * 16 + w]; }
    printf("stream %-26s %s: %6.1f cycles per 128 instructions of one wavefront | per wavefront 0..7 (to the barrier): ", name, BAR ? "barrier / 128" : "free-running ",
           tot / 256 / iters);
    for (int w = 0; w < 8; ++w) printf("%5.0f ", arr[w] / 256 / iters);
    printf("\n");
}

int main() {
    float *W, *Y; unsigned long long *cyc;
    (void)hipMalloc(&W, 128 * 128 * 4); (void)hipMalloc(&Y, 256 * 512 * 16); (void)hipMalloc(&cyc, 256 * 16 * 8);
    static float hw[128 * 128];
    for (int i = 0; i < 128 * 128; ++i) hw[i] = 0.05f * ((i * 37 % 101) - 50) / 50.f;
    (void)hipMemcpy(W, hw, sizeof(hw), hipMemcpyHostToDevice);
    const int reps = 2000;
#define BOTH(P, NAME) run<0, P>(W, Y, cyc, reps, NAME); run<1, P>(W, Y, cyc, reps, NAME);
    BOTH(0, "no priority")
    BOTH(1, "wavefronts 4-7 prio 1")
    BOTH(6, "wavefronts 4-7 prio 3")
    BOTH(2, "wavefronts 0-3 prio 1")
    BOTH(3, "prio 1 alternating by round")
    BOTH(4, "wavefronts 4-7 start 64 cycles late")
    BOTH(5, "wavefronts 4-7 prio 1 during GELU")
    BOTH(7, "wavefronts 4-7 start 192 cycles late")
    BOTH(8, "wavefronts 4-7 start 384 cycles late")
    BOTH(9, "wavefronts 4-7: epilogue per row block")
    BOTH(10, "GELU with scalar v_fma_f32")
    float *sink; (void)hipMalloc(&sink, 4096);
    run_stream<0, 0>(cyc, sink, "v_pk_fma_f32 x 8 regs"); run_stream<0, 1>(cyc, sink, "v_pk_fma_f32 x 8 regs");
    run_stream<1, 0>(cyc, sink, "v_fma_f32 x 8 regs"); run_stream<1, 1>(cyc, sink, "v_fma_f32 x 8 regs");
    run_stream<2, 0>(cyc, sink, "7 pk_fma + 1 v_exp"); run_stream<2, 1>(cyc, sink, "7 pk_fma + 1 v_exp");
    run_stream<3, 0>(cyc, sink, "pk_fma, 2 dependent chains"); run_stream<3, 1>(cyc, sink, "pk_fma, 2 dependent chains");
    return 0;
}
