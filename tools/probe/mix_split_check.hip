// Is (v_fma_mixlo_f16, v_fma_mixhi_f16) of x - h the same 16 bits as v_cvt_pk_f16_f32(x - float(h))? (SplitH2::split2)
// hipcc --offload-arch=gfx950 -O3 tools/probe/mix_split_check.hip -o tools/probe/mix_split_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, unsigned *o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f2 v = f2{x[2 * i], x[2 * i + 1]};
    const h2 h = __builtin_convertvector(v, h2);
    const f2 r = v - __builtin_convertvector(h, f2);
    const h2 lref = __builtin_convertvector(r, h2);
    unsigned hb = __builtin_bit_cast(unsigned, h), l;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v.x), "v"(hb));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v.y), "v"(hb));
    o[2 * i] = __builtin_bit_cast(unsigned, lref);
    o[2 * i + 1] = l;
}
int main() {
    const int n = 1 << 22;
    float *hx = (float *)malloc(2 * n * 4);
    unsigned *ho = (unsigned *)malloc(2 * n * 4);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) {
        unsigned b = ((unsigned)rand() << 16) ^ (unsigned)rand();
        if (i % 3 == 0) { float f = (float)((rand() % 200001) - 100000) * 1e-4f; memcpy(&b, &f, 4); }   // O(1..10) values
        if (i % 7 == 0) { float f = (float)((rand() % 2001) - 1000) * 1e-9f; memcpy(&b, &f, 4); }      // tiny values
        unsigned e = (b >> 23) & 255;
        if (e == 255) b &= 0x807fffffu;                      // no inf / nan inputs
        memcpy(&hx[i], &b, 4);
    }
    float *dx; unsigned *dout;
    hipMalloc(&dx, 2 * n * 4); hipMalloc(&dout, 2 * n * 4);
    hipMemcpy(dx, hx, 2 * n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(ho, dout, 2 * n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_finite = 0;
    for (int i = 0; i < n; ++i)
        if (ho[2 * i] != ho[2 * i + 1]) {
            ++bad;
            const float a = hx[2 * i], b = hx[2 * i + 1];
            if (fabsf(a) < 65504.f && fabsf(b) < 65504.f) { if (bad_finite++ < 5) printf("x = %g %g ref %08x mix %08x\n", a, b, ho[2 * i], ho[2 * i + 1]); }
        }
    printf("pairs %d, mismatches %ld (with both |x| < 65504: %ld)\n", n, bad, bad_finite);
    return 0;
}
