#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
// cycles per wave-instruction on one SIMD: one wave per SIMD (256 threads per workgroup, one workgroup per CU), 8 independent chains
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t v[8]; unsigned long long w[8]; float f[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 2654435761u + i + seed; w[i] = v[i]; f[i] = 1.0f + 1e-3f * (float)((threadIdx.x + i) & 15); }
    const uint32_t M = 0xD2511F53u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (OP == 0) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(v[i]), "s"(M) : "vcc"); v[i] = (uint32_t)(w[i] >> 32) ;
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "s"(M));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "s"(M));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(v[i]) : "v"(v[i]), "v"(v[(i + 1) & 7]), "s"(M));
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(i) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(v[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(i) asm volatile("v_exp_f32 %0, %1" : "=v"(f[i]) : "v"(f[i]));
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(f[i]) : "v"(f[i]), "s"(0.999f));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(w[i]) : "v"(w[i]));
                REP8(X)
#undef X
            } else if (OP == 8) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %1 op_sel_hi:[1,0,0]" : "=v"(v[i]) : "v"(v[i]), "s"(0.5f));
                REP8(X)
#undef X
            } else if (OP == 9) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(v[i]) : "v"(f[i]));
                REP8(X)
#undef X
            } else if (OP == 10) {
#define X(i) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "s"(M));
                REP8(X)
#undef X
            } else if (OP == 11) {
#define X(i) asm volatile("v_sin_f32 %0, %1" : "=v"(f[i]) : "v"(f[i]));
                REP8(X)
#undef X
            } else if (OP == 12) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(v[i]) : "v"(v[i]), "s"(M));
                REP8(X)
#undef X
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= v[i] ^ (uint32_t)w[i] ^ __float_as_uint(f[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, uint32_t* out) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, 100, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 32;
    printf("%-20s %8.3f ms  %6.2f ns per wave-instruction  (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 256 * 4);
    run<6>("v_fma_f32", out); run<6>("v_fma_f32", out);
    run<0>("v_mad_u64_u32", out); run<1>("v_mul_lo_u32", out); run<2>("v_mul_hi_u32", out); run<10>("v_mul_u32_u24", out); run<12>("v_mad_u32_u24", out);
    run<3>("v_bitop3_b32", out); run<4>("v_xor_b32", out); run<5>("v_exp_f32", out); run<11>("v_sin_f32", out);
    run<7>("v_pk_fma_f32", out); run<8>("v_fma_mixlo_f16", out); run<9>("v_cvt_pk_f16_f32", out);
    return 0;
}
