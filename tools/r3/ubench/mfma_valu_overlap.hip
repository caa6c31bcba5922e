// Does a single wave overlap its own VALU work with its v_mfma_f32_16x16x32_f16 stream?  One wave per SIMD (256 threads per
// workgroup, one workgroup per CU): MFMA only / VALU only / interleaved 1 MFMA : NV VALU.  Round-3 planning probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x ^ e)); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + 1e-3f * (float)((threadIdx.x + i) & 15);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE != 1) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[r & 3], 0, 0, 0);
            if (MODE != 0) {
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(r * NV + j) & 7]) : "s"(0.999f));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int NV>
float run(float* out, int waves) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256 * waves), dim3(256), 0, 0, out, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256 * waves), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / (iters * 8.0f);   // ns per (MFMA [+ NV VALU]) group
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 2 * 256 * 4);
    for (int waves = 1; waves <= 2; ++waves) {
        printf("--- %d wave(s) per SIMD\n", waves);
        printf("MFMA only:                %.2f ns per MFMA\n", run<0, 0>(out, waves));
        printf("VALU only, 3 per group:   %.2f ns per group\n", run<1, 3>(out, waves));
        printf("1 MFMA + 3 VALU:          %.2f ns per group\n", run<2, 3>(out, waves));
        printf("VALU only, 6 per group:   %.2f ns per group\n", run<1, 6>(out, waves));
        printf("1 MFMA + 6 VALU:          %.2f ns per group\n", run<2, 6>(out, waves));
        printf("1 MFMA + 2 VALU:          %.2f ns per group\n", run<2, 2>(out, waves));
    }
    return 0;
}
