// Issue cost of the VALU instruction classes the per-edge kernels' epilogues are made of, measured the way those kernels run them:
// 8 wavefronts per workgroup = 2 per SIMD, every wavefront a dense stream of INDEPENDENT instructions of one class (8 destination
// registers in rotation). Output: cycles per instruction as seen by one SIMD (two wavefronts issuing) and by one wavefront alone
// (wavefronts 4-7 idle). These are the `c` of bench.py's issue-roof bracket (round 5).
// hipcc --offload-arch=gfx950 -O3 tools/probe/valu_cost_probe.hip -o tools/probe/valu_cost_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

// one asm statement = 8 instructions of one class on 8 different registers (separate statements get an s_nop between them)
#define A8 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define P8 "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
#define I8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define S_FMA(k)   "v_fma_f32 %" #k ", %8, %9, %" #k "\n\t"
#define S_PKFMA(k) "v_pk_fma_f32 %" #k ", %8, %9, %" #k "\n\t"
#define S_PKADD(k) "v_pk_add_f32 %" #k ", %8, %" #k "\n\t"
#define S_EXP(k)   "v_exp_f32 %" #k ", %" #k "\n\t"
#define S_RSQ(k)   "v_rsq_f32 %" #k ", %" #k "\n\t"
#define S_MIN3(k)  "v_minimum3_f32 %" #k ", |%" #k "|, %9, %9\n\t"
#define S_CVTPK(k) "v_cvt_pk_f16_f32 %" #k ", %" #k ", %9\n\t"
#define S_CVTLO(k) "v_cvt_f32_f16 %" #k ", %" #k "\n\t"
#define S_CVTHI(k) "v_cvt_f32_f16_sdwa %" #k ", %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
#define S_MOV(k)   "v_mov_b32 %" #k ", %" #k "\n\t"
#define S_DPP(k)   "v_add_f32_dpp %" #k ", %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define S_CND(k)   "v_cndmask_b32 %" #k ", %" #k ", %9, vcc\n\t"
#define S_MIXLO(k) "v_fma_mixlo_f16 %" #k ", %8, 1.0, -%" #k " op_sel_hi:[0,0,1]\n\t"
#define S_SWAP(k)  "v_permlane16_swap_b32 %" #k ", %" #k "\n\t"
#define S_PKFMAS(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %10 op_sel_hi:[1,1,0]\n\t"
#define S_FMAS(k)  "v_fma_f32 %" #k ", %" #k ", %8, %11\n\t"
#define S_CNDS(k)  "v_cndmask_b32_e64 %" #k ", %" #k ", %9, %10\n\t"
#define S_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n\t"
#define S_MAX3(k)  "v_maximum3_f32 %" #k ", %" #k ", 0, 0\n\t"
#define S_CNDV(k)  "v_cndmask_b32_e64 %" #k ", %" #k ", 0, vcc\n\t"
#define S_ADD(k)   "v_add_f32 %" #k ", %" #k ", %8\n\t"
#define S_MUL(k)   "v_mul_f32 %" #k ", %" #k ", %8\n\t"
#define S_MULHI(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n\t"
#define S_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n\t"
#define S_ADD3(k)  "v_add3_u32 %" #k ", %" #k ", %8, %9\n\t"
#define S_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 3, %8\n\t"
#define S_ADDU(k)  "v_add_u32 %" #k ", %" #k ", %8\n\t"
#define OPA(S) asm volatile(I8(S) : A8 : "v"(x), "v"(y), "s"(sc), "s"(sf));
#define OPP(S) asm volatile(I8(S) : P8 : "v"(px), "v"(py), "s"(sc), "s"(sf));

typedef float f2 __attribute__((ext_vector_type(2)));

template <int CLS>
__global__ __launch_bounds__(512, 1) void cost_kernel(int iters, int waves, unsigned long long *out, float *sink) {
    const int wv = threadIdx.x >> 6;
    float a[8], b[8];
    f2 p[8];
    const float x = 1.0001f + 1e-6f * threadIdx.x, y = 1e-7f * (threadIdx.x + 1);
    const f2 px = f2{x, x}, py = f2{y, y};
    const unsigned long long sc = 0x3f8000013f800001ull + (unsigned long long)iters;   // an SGPR pair (constant / lane mask)
    const float sf = 1.0f + 1e-6f * iters;
    for (int k = 0; k < 8; ++k) { a[k] = 1e-3f * (threadIdx.x + k); b[k] = 0.5f + 1e-3f * k; p[k] = f2{a[k], b[k]}; }
    unsigned long long c0 = 0, c1 = 0;
    if (wv < waves) {
        c0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#define BODY(M, S) M(S) M(S) M(S) M(S)
            if (CLS == 0) { BODY(OPA, S_FMA) }
            if (CLS == 1) { BODY(OPP, S_PKFMA) }
            if (CLS == 2) { BODY(OPP, S_PKADD) }
            if (CLS == 3) { BODY(OPA, S_EXP) }
            if (CLS == 4) { BODY(OPA, S_RSQ) }
            if (CLS == 5) { BODY(OPA, S_MIN3) }
            if (CLS == 6) { BODY(OPA, S_CVTPK) }
            if (CLS == 7) { BODY(OPA, S_CVTLO) }
            if (CLS == 8) { BODY(OPA, S_CVTHI) }
            if (CLS == 9) { BODY(OPA, S_MOV) }
            if (CLS == 10) { BODY(OPA, S_DPP) }
            if (CLS == 11) { BODY(OPA, S_CND) }
            if (CLS == 12) { BODY(OPA, S_MIXLO) }
            if (CLS == 13) { BODY(OPA, S_SWAP) }
            if (CLS == 14) { BODY(OPP, S_PKFMAS) }
            if (CLS == 15) { BODY(OPA, S_FMAS) }
            if (CLS == 16) { BODY(OPA, S_CNDS) }
            if (CLS == 17) { BODY(OPP, S_PKMUL) }
            if (CLS == 18) { BODY(OPA, S_MAX3) }
            if (CLS == 19) { BODY(OPA, S_CNDV) }
            if (CLS == 20) { BODY(OPA, S_ADD) }
            if (CLS == 21) { BODY(OPA, S_MUL) }
            if (CLS == 22) { BODY(OPA, S_MULHI) }
            if (CLS == 23) { BODY(OPA, S_MULLO) }
            if (CLS == 24) { BODY(OPA, S_ADD3) }
            if (CLS == 25) { BODY(OPA, S_LSHLOR) }
            if (CLS == 26) { BODY(OPA, S_ADDU) }
        }
        c1 = __builtin_readcyclecounter();
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += a[k] + b[k] + p[k].x + p[k].y;
    if (s == 123.456f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wv] = c1 - c0;
}

// Dependent-issue latency: NCH independent chains of v_pk_fma_f32 (q <- q t + c) in round-robin order, each instruction depending on
// the one NCH places before it — the GELU's Horner chains. NCH = 1: a pure dependent chain.
template <int NCH, int PK>
__global__ __launch_bounds__(512, 1) void chain_kernel(int iters, int waves, unsigned long long *out, float *sink) {
    const int wv = threadIdx.x >> 6;
    f2 q[NCH];
    const float x = 0.9999f + 1e-7f * threadIdx.x;
    const f2 t = f2{x, x}, c = f2{1e-7f, 1e-7f};
    for (int k = 0; k < NCH; ++k) q[k] = f2{1e-3f * (threadIdx.x + k), 0.5f + 1e-3f * k};
    unsigned long long c0 = 0, c1 = 0;
    if (wv < waves) {
        c0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 48 / NCH; ++r)
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[k]) : "v"(t), "v"(c));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[k].x) : "v"(t.x), "v"(c.x));
                }
        }
        c1 = __builtin_readcyclecounter();
    }
    float s = 0.f;
    for (int k = 0; k < NCH; ++k) s += q[k].x + q[k].y;
    if (s == 123.456f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wv] = c1 - c0;
}
template <int NCH, int PK>
void run_chain() {
    unsigned long long *out; float *sink;
    (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&sink, 4096);
    const int iters = 400;
    double res[2];
    for (int v = 0; v < 2; ++v) {
        const int waves = v ? 4 : 8;
        chain_kernel<NCH, PK><<<256, 512>>>(iters, waves, out, sink);
        chain_kernel<NCH, PK><<<256, 512>>>(iters, waves, out, sink);
        (void)hipDeviceSynchronize();
        static unsigned long long h[2048];
        (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double c = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) c += h[b * 8 + w];
        c /= 256.0 * waves;
        res[v] = c / (iters * (48 / NCH) * NCH);
    }
    printf("%s, %d chain(s) round-robin:  two wavefronts per SIMD %6.2f cycles per instruction per SIMD (%6.2f per wavefront)   one wavefront per SIMD %6.2f\n",
           PK ? "v_pk_fma_f32" : "v_fma_f32   ", NCH, res[0] / 2.0, res[0], res[1]);
    (void)hipFree(out); (void)hipFree(sink);
}

template <int CLS>
void run(const char *name) {
    unsigned long long *out; float *sink;
    (void)hipMalloc(&out, 256 * 8 * 8); (void)hipMalloc(&sink, 4096);
    const int iters = 500;
    double res[2];
    for (int v = 0; v < 2; ++v) {
        const int waves = v ? 4 : 8;
        cost_kernel<CLS><<<256, 512>>>(iters, waves, out, sink);
        cost_kernel<CLS><<<256, 512>>>(iters, waves, out, sink);
        (void)hipDeviceSynchronize();
        static unsigned long long h[2048];
        (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double c = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) c += h[b * 8 + w];
        c /= 256.0 * waves;
        res[v] = c / (iters * 32.0);
    }
    // two wavefronts per SIMD issue 2 instructions in the time one wavefront measures: per-SIMD cost = wavefront cost / 2
    printf("%-28s two wavefronts per SIMD: %6.2f cycles per instruction per SIMD (%6.2f seen by each wavefront)   one wavefront per SIMD: %6.2f\n",
           name, res[0] / 2.0, res[0], res[1]);
    (void)hipFree(out); (void)hipFree(sink);
}
int main() {
    run<0>("v_fma_f32");
    run<1>("v_pk_fma_f32");
    run<2>("v_pk_add_f32");
    run<3>("v_exp_f32");
    run<4>("v_rsq_f32");
    run<5>("v_minimum3_f32");
    run<6>("v_cvt_pk_f16_f32");
    run<7>("v_cvt_f32_f16");
    run<8>("v_cvt_f32_f16_sdwa (hi half)");
    run<9>("v_mov_b32");
    run<10>("v_add_f32_dpp quad_perm");
    run<11>("v_cndmask_b32");
    run<12>("v_fma_mixlo_f16");
    run<13>("v_permlane16_swap_b32");
    run<14>("v_pk_fma_f32, SGPR-pair src2");
    run<15>("v_fma_f32, SGPR src2");
    run<16>("v_cndmask_b32_e64, SGPR mask");
    run<17>("v_pk_mul_f32");
    run<18>("v_maximum3_f32 x, 0, 0");
    run<19>("v_cndmask_b32_e64 x, 0, vcc");
    run<20>("v_add_f32");
    run<21>("v_mul_f32");
    run<22>("v_mul_hi_u32");
    run<23>("v_mul_lo_u32");
    run<24>("v_add3_u32");
    run<25>("v_lshl_or_b32");
    run<26>("v_add_u32");
    run_chain<1, 1>(); run_chain<2, 1>(); run_chain<3, 1>(); run_chain<4, 1>(); run_chain<6, 1>(); run_chain<8, 1>();
    run_chain<1, 0>(); run_chain<2, 0>(); run_chain<3, 0>(); run_chain<4, 0>(); run_chain<6, 0>();
    return 0;
}
