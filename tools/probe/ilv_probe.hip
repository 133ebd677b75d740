// Can the GELU + split epilogue of one 16-row block ride in the MFMA shadow of the NEXT row block's GEMM steps, in the SAME wavefront?
// (round 5; the per-edge kernels run all eight wavefronts GEMM -> epilogue -> barrier in lock step: the matrix pipe idles through
// every epilogue, the VALU through every GEMM.)
// One workgroup of 8 wavefronts per CU runs R x { tile GEMM 48 x 128 x 128 (f16x2, 36 MFMAs per wavefront) from plane tile t,
// GELU + split of the result into plane tile t ^ 1, ONE barrier } — the shape of GEMM 1 / GEMM 2 of the edge update.
//   FORM 0  shipped order: k-major steps (row blocks cycle), all MFMAs, then the three epilogues
//   FORM 1  row-block-major steps; the epilogue of row block rb - 1 is cut into 4 chunks, chunk k sits behind the three MFMAs of
//           step (rb, k); scheduling barriers between steps, an MFMA : VALU group pattern inside (ONE accumulator chain per row block)
//   FORM 2  as 1 with TWO accumulators per row block (even / odd k) summed in the epilogue: no VALU ever sits between two MFMAs
//           that touch the same accumulator
//   FORM 3  as 1 without the group pattern (the compiler orders each step's MFMAs + chunk as it likes)
//   FORM 4  row-block-major, epilogue of rb - 1 as ONE block in front of the MFMAs of row block rb in program order, no pins:
//           only the two wavefronts of a SIMD drifting apart can overlap anything
//   FORM 5/6 = GEMM only / epilogue only (the parts), 7/8 = the same without the barrier, 9 = the barrier alone
//   FORM 10 / 11 = shipped GEMM, epilogue one row block at a time (GELU -> split -> plane write) / with the write one row block late
// ASM = 1: the packed Horner steps as inline asm (what the shipped kernels use; invisible to the group pattern); 0: builtin fma
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -DTM_GELU_NAN3=1 -DILV_ASM=$ASM -I thermompnn_amd/csrc
//       tools/probe/ilv_probe.hip -o tools/probe/ilv_probe_asm$ASM
#include <stdio.h>

#include <type_traits>

#include "tmpnn_split.h"

#ifndef ILV_ASM
#define ILV_ASM 1           // 1: the packed Horner steps as inline asm (pk_horner of tmpnn_common.h, what the shipped kernels run); 0: builtin fma
#endif
#ifndef ILV_VPER
#define ILV_VPER 5          // VALU / transcendental instructions requested behind each MFMA of a step (FORM 1 / 2)
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

#define TM_HF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0)

// gelu2 (tmpnn_common.h, TM_GELU_NAN3 form) cut in two: the same operations in the same order, so the same bits
__device__ __forceinline__ f2 pkh(f2 q, f2 t, float c) {
#if ILV_ASM
    return pk_horner(q, t, c);
#else
    return __builtin_elementwise_fma(q, t, f2{c, c});
#endif
}
struct GeluMid { f2 t, q; };
__device__ __forceinline__ GeluMid gelu2_head(f2 x) {
    GeluMid g;
    g.t = f2{__builtin_elementwise_minimum(fabsf(x.x), 5.656854249f), __builtin_elementwise_minimum(fabsf(x.y), 5.656854249f)};
    f2 q = __builtin_elementwise_fma(f2{3.309543916e-05f, 3.309543916e-05f}, g.t, f2{-7.692427171e-04f, -7.692427171e-04f});
    q = pkh(q, g.t, 8.080792133e-03f);
    q = pkh(q, g.t, -5.341222090e-02f);
    g.q = pkh(q, g.t, -4.587708865e-01f);
    return g;
}
__device__ __forceinline__ f2 gelu2_tail(const GeluMid g, f2 x) {
    f2 q = pkh(g.q, g.t, -1.151201730e+00f);
    const f2 e = pkh(q, g.t, -9.999930581e-01f);
    return __builtin_elementwise_fma(-g.t, f2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)},
                                     f2{__builtin_elementwise_maximum(x.x, 0.f), __builtin_elementwise_maximum(x.y, 0.f)});
}
// chunk K of the epilogue of one row block: 0 = the head of both GELU pairs (clamp + 4 Horner steps), 1 = their tails (2 steps, exp,
// max, last fma), 2 = both splits, 3 = the two 8-byte plane writes
struct EpState {
    GeluMid m01, m23;
    f2 g01, g23;
    unsigned a[2], b[2];
};
template <int K>
__device__ __forceinline__ void ep_chunk(EpState &s, const f4 v, char *dst, int row, int c4) {
    if constexpr (K == 0) { s.m01 = gelu2_head(f2{v.x, v.y}); s.m23 = gelu2_head(f2{v.z, v.w}); }
    if constexpr (K == 1) { s.g01 = gelu2_tail(s.m01, f2{v.x, v.y}); s.g23 = gelu2_tail(s.m23, f2{v.z, v.w}); }
    if constexpr (K == 2) {
        SplitH2::split2(s.g01, s.a);
        SplitH2::split2(s.g23, s.b);
    }
    if constexpr (K == 3) {
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<u2 *>(dst + plane_off4<>(p, row, c4)) = u2{s.a[p], s.b[p]};
    }
}

template <int FORM>
__device__ __forceinline__ f4 round_rbmajor(const char *src, char *dst, const WFragS<SplitH2> (&w)[1][4], f4 init, int lane, int c4) {
    constexpr int PF = 2, NS = 12, NB = PF + 1;
    const int m = lane & 15, q = lane >> 4;
    f4 acc[3], acc2[3];
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) { acc[rb] = init * (1.0f + 0.25f * rb); acc2[rb] = f4{0.f, 0.f, 0.f, 0.f}; }
    u4 x[NB][2];
    auto rd = [&](int s) {          // step s = 4 rb + k
#pragma unroll
        for (int p = 0; p < 2; ++p) x[s % NB][p] = *reinterpret_cast<const u4 *>(src + plane_off8<>(p, 16 * (s / 4) + m, 4 * (s % 4) + q));
    };
    EpState es;
    f4 keep = f4{0.f, 0.f, 0.f, 0.f};
    sfor<0, PF>([&](auto S) { rd(decltype(S)::value); });
    sfor<0, NS>([&](auto S) {
        constexpr int s = decltype(S)::value, rb = s / 4, k = s % 4;
        if constexpr (s + PF < NS) rd(s + PF);
        if constexpr (FORM != 4) __builtin_amdgcn_sched_barrier(0);
        if constexpr (FORM == 4 && k == 0 && rb > 0) {
            sfor<0, 4>([&](auto K) { ep_chunk<decltype(K)::value>(es, FORM == 2 ? acc[rb - 1] + acc2[rb - 1] : acc[rb - 1], dst, 16 * (rb - 1) + m, c4); });
        }
        f4 &a = (FORM == 2 && (k & 1)) ? acc2[rb] : acc[rb];
        a = TM_HF(w[0][k].p[1], x[s % NB][0], a);
        a = TM_HF(w[0][k].p[0], x[s % NB][1], a);
        a = TM_HF(w[0][k].p[0], x[s % NB][0], a);
        if constexpr (FORM != 4 && rb > 0) {
            ep_chunk<k>(es, FORM == 2 ? acc[rb - 1] + acc2[rb - 1] : acc[rb - 1], dst, 16 * (rb - 1) + m, c4);
            if constexpr (FORM == 1 || FORM == 2) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x402, ILV_VPER, 0);
                }
            }
        }
        if constexpr (FORM != 4) __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, 4>([&](auto K) { ep_chunk<decltype(K)::value>(es, FORM == 2 ? acc[2] + acc2[2] : acc[2], dst, 32 + m, c4); });
    keep = f4{es.g01.x, es.g01.y, es.g23.x, es.g23.y};
    return keep;
}

template <int FORM>
__global__ __launch_bounds__(512, 2) void ilv_kernel(const float *__restrict__ W, float *__restrict__ Y, int reps, unsigned long long *cyc) {
    using SP = SplitH2;
    __shared__ __attribute__((aligned(16))) char tA[2][2 * SPLIT_PLANE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int c4 = 4 * wv + q;
    WFragS<SP> w[1][4];
    load_wfrag_split<SP, 4>(W, TM_H, 16 * wv, 0, TM_H, w[0], lane);
    for (int t = 0; t < 2; ++t)
        for (int rb = 0; rb < 3; ++rb) {
            const f4 v = f4{0.01f * (lane + rb), -0.02f * (wv + t), 0.003f * m, 0.5f - 0.01f * q};
            store_split<SP>(tA[t], 16 * rb + m, c4, v);
        }
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    f4 keep = f4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        const int t = r & 1;
        f4 init = f4{0.1f, 0.2f, 0.3f, 0.4f} + keep * 1e-3f;
        touch(init);
        if constexpr (FORM == 0 || FORM == 5 || FORM == 6 || FORM >= 7) {
            f4 acc[3][1];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = init * (1.0f + 0.25f * rb);
            if (FORM != 6 && FORM != 8 && FORM != 9) mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, 2>(tA[t], w, acc, lane);
            __builtin_amdgcn_sched_barrier(0);
            if (FORM == 9) keep = acc[0][0];
            else if (FORM == 10 || FORM == 11) {
                // one row block at a time: GELU -> split -> plane write, so that the LDS writes of row blocks 0 and 1 drain under the
                // VALU work that follows them (FORM 0 issues all three writes at the very end, in front of the barrier); FORM 11 also
                // delays each write by one row block (it is issued behind the NEXT row block's GELU)
                f4 g[3];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    g[rb] = gelu4(acc[rb][0]);
                    if (FORM == 10) store_split<SP>(tA[t ^ 1], 16 * rb + m, c4, g[rb]);
                    if (FORM == 11 && rb > 0) store_split<SP>(tA[t ^ 1], 16 * (rb - 1) + m, c4, g[rb - 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (FORM == 11) store_split<SP>(tA[t ^ 1], 32 + m, c4, g[2]);
                keep = g[2];
            } else if (FORM != 5 && FORM != 7) {
                f4 g[3];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) store_split<SP>(tA[t ^ 1], 16 * rb + m, c4, g[rb]);
                keep = g[2];
            } else keep = acc[0][0] + acc[1][0] + acc[2][0];
        } else {
            keep = round_rbmajor<FORM>(tA[t], tA[t ^ 1], w, init, lane, c4);
        }
        if (FORM != 7 && FORM != 8) __syncthreads();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    st4(Y + ((size_t)blockIdx.x * 512 + tid) * 4, keep);
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int FORM>
double run(const float *W, float *Y, unsigned long long *cyc, int reps, double *chk) {
    ilv_kernel<FORM><<<256, 512>>>(W, Y, reps, cyc);
    ilv_kernel<FORM><<<256, 512>>>(W, Y, reps, cyc);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    static float y[512 * 4];
    (void)hipMemcpy(y, Y, sizeof(y), hipMemcpyDeviceToHost);
    double s = 0, c = 0;
    for (int i = 0; i < 256; ++i) s += h[i];
    for (int i = 0; i < 512 * 4; ++i) c += y[i];
    *chk = c;
    return s / 256 / reps;
}

int main() {
    float *W, *Y; unsigned long long *cyc;
    (void)hipMalloc(&W, 128 * 128 * 4); (void)hipMalloc(&Y, 256 * 512 * 16); (void)hipMalloc(&cyc, 264 * 8);
    static float hw[128 * 128];
    for (int i = 0; i < 128 * 128; ++i) hw[i] = 0.05f * ((i * 37 % 101) - 50) / 50.f;
    (void)hipMemcpy(W, hw, sizeof(hw), hipMemcpyHostToDevice);
    const int reps = 2001;
    double c[12], k[12];
    k[0] = run<0>(W, Y, cyc, reps, &c[0]);
    k[1] = run<1>(W, Y, cyc, reps, &c[1]);
    k[2] = run<2>(W, Y, cyc, reps, &c[2]);
    k[3] = run<3>(W, Y, cyc, reps, &c[3]);
    k[4] = run<4>(W, Y, cyc, reps, &c[4]);
    k[5] = run<5>(W, Y, cyc, reps, &c[5]);
    k[6] = run<6>(W, Y, cyc, reps, &c[6]);
    k[7] = run<7>(W, Y, cyc, reps, &c[7]);
    k[8] = run<8>(W, Y, cyc, reps, &c[8]);
    k[9] = run<9>(W, Y, cyc, reps, &c[9]);
    k[10] = run<10>(W, Y, cyc, reps, &c[10]);
    k[11] = run<11>(W, Y, cyc, reps, &c[11]);
    printf("ASM %d VPER %d: cycles per round (GEMM + GELU/split + barrier), one workgroup per CU:\n", ILV_ASM, ILV_VPER);
    const char *nm[12] = {"0 shipped order", "1 chunks, grouped", "2 chunks, 2 accs", "3 chunks, free", "4 block, no pins", "5 GEMM only", "6 epilogue only",
                          "7 GEMM, no barrier", "8 epilogue, no barrier", "9 barrier only", "10 epilogue per row block", "11 ... writes one block late"};
    for (int i = 0; i < 12; ++i) printf("   form %-20s %7.0f cycles  (%+5.1f %% vs shipped)   checksum %.6e\n", nm[i], k[i], 100.0 * (k[i] - k[0]) / k[0], c[i]);
    return 0;
}
