// Developer micro-benchmark: SIMD cycles per wave64 VALU instruction on gfx950 for the instruction kinds the FFT
// kernels are made of, at 1, 2 and 4 waves per SIMD.  16 independent chains per lane, so latency is not the limit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int KIND>
__global__ void rate(double* out, int iters, double seed) {
    double d[16]; float f[16]; int n[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { d[i] = seed + i + threadIdx.x; f[i] = (float)d[i]; n[i] = (int)d[i]; }
    const double c = seed * 1.0000001, c2 = seed * 0.5; const float cf = (float)c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(c), "v"(c2));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf));
                if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
                if (KIND == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));
                if (KIND == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
                if (KIND == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                if (KIND == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(c));
                if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(n[(i + 1) & 15]));
                // round 3: the forms a complex multiply can be made of
                if (KIND == 10) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
                if (KIND == 11) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[i]) : "v"(cf), "v"(f[(i + 1) & 15]));
                if (KIND == 12) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                if (KIND == 13) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                if (KIND == 14) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(cf), "v"(f[(i + 1) & 15]));
                if (KIND == 15) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(d[i]) : "v"(c), "v"(d[(i + 1) & 15]));
                if (KIND == 16) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += d[i] + f[i] + n[i];
    if (s == 1.2345) out[0] = s;
}

template <int KIND>
void run(const char* name, double* out) {
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4}) {           // waves per SIMD: block = 256 * wps threads, one block per CU
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, 100, 1.0);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)iters * 64 * wps;           // wave-instructions issued on each SIMD
        printf("%-14s waves/SIMD=%d : %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}

int main() {
    double* out; CK(hipMalloc(&out, 64));
    run<0>("v_add_f64", out); run<1>("v_mul_f64", out); run<2>("v_fma_f64", out);
    run<3>("v_fma_f32", out); run<4>("v_add_f32", out); run<5>("v_add_u32", out);
    run<6>("v_cvt_f32_f64", out); run<7>("v_cvt_f64_f32", out); run<8>("v_pk_fma_f32", out); run<9>("v_mov_b32", out);
    run<10>("v_mul_f32_e32", out); run<11>("v_fmac_f32_e32", out); run<12>("v_pk_mul_f32", out); run<13>("v_pk_add_f32", out);
    run<14>("v_fma_f32 3src", out); run<15>("v_pk_fma opsel", out); run<16>("v_sub_f32 2vgpr", out);
    return 0;
}
