#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
#define BODY(NOP) \
    asm volatile( \
        "v_mov_b32 v44, %[a0]\n v_mov_b32 v45, %[a1]\n v_mov_b32 v46, %[a2]\n v_mov_b32 v47, %[a3]\n s_nop 7\n" \
        "v_fma_mixlo_f16 v40, %[h0], %[s], %[t0] op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 v40, %[h0], %[s], %[t1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
        "v_fma_mixlo_f16 v41, %[h1], %[s], %[t2] op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 v41, %[h1], %[s], %[t3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
        "v_fma_mixlo_f16 v42, %[h2], %[s], %[t4] op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 v42, %[h2], %[s], %[t5] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
        "v_fma_mixlo_f16 v43, %[h3], %[s], %[t6] op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 v43, %[h3], %[s], %[t7] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
        NOP \
        "v_mfma_f32_16x16x32_f16 v[48:51], v[44:47], v[40:43], 0\n s_nop 15\n s_nop 15\n" \
        "v_mov_b32 %[o0], v48\n v_mov_b32 %[o1], v49\n v_mov_b32 %[o2], v50\n v_mov_b32 %[o3], v51\n" \
        : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3) \
        : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [h0] "v"(h0), [h1] "v"(h1), [h2] "v"(h2), [h3] "v"(h3), [s] "s"(-2048.f), \
          [t0] "v"(t[0]), [t1] "v"(t[1]), [t2] "v"(t[2]), [t3] "v"(t[3]), [t4] "v"(t[4]), [t5] "v"(t[5]), [t6] "v"(t[6]), [t7] "v"(t[7]) \
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51")
template <int V>
__global__ void k(const float* in, float* out) {
    float x[8], t[8];
    for (int e = 0; e < 8; ++e) { x[e] = in[8 * threadIdx.x + e]; t[e] = x[e] * 2048.f; }
    const uint32_t a0 = 0x3c003e00u + threadIdx.x, a1 = 0x40003c00u, a2 = 0x3c004200u, a3 = 0x3e003c00u ^ (threadIdx.x << 3);
    const uint32_t h0 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2_t){x[0], x[1]}, h2_t));
    const uint32_t h1 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2_t){x[2], x[3]}, h2_t));
    const uint32_t h2 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2_t){x[4], x[5]}, h2_t));
    const uint32_t h3 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2_t){x[6], x[7]}, h2_t));
    float o0, o1, o2, o3;
    if (V == 0) BODY("");
    else if (V == 2) BODY("s_nop 0\n");
    else if (V == 3) BODY("s_nop 1\n");
    else if (V == 4) BODY("s_nop 3\n");
    else if (V == 5) BODY("v_mov_b32 v43, v43\n");
    else if (V == 6) BODY("v_mov_b32 v43, v43\n s_nop 0\n");
    else if (V == 7) BODY("v_mov_b32 v43, v43\n s_nop 1\n");
    else BODY("s_nop 7\n s_nop 7\n");
    out[4 * threadIdx.x] = o0; out[4 * threadIdx.x + 1] = o1; out[4 * threadIdx.x + 2] = o2; out[4 * threadIdx.x + 3] = o3;
}
int main() {
    const int NT = 64;
    std::vector<float> in(8 * NT), r0(4 * NT), r1(4 * NT);
    float *din, *o0, *o1;
    (void)hipMalloc(&din, 8 * NT * 4); (void)hipMalloc(&o0, 4 * NT * 4); (void)hipMalloc(&o1, 4 * NT * 4);
    srand(3);
    long bad = 0, tot = 0; long badv[7] = {0,0,0,0,0,0,0}; (void)bad;
    for (int rep = 0; rep < 3000; ++rep) {
        for (auto& f : in) f = ((rand() % 2000001) - 1000000) * 1e-6f * 3.f;
        (void)hipMemcpy(din, in.data(), 8 * NT * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(NT), 0, 0, din, o1);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(r1.data(), o1, 4 * NT * 4, hipMemcpyDeviceToHost);
        void (*ks[7])(const float*, float*) = {k<0>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>};
        for (int v = 0; v < 7; ++v) {
            hipLaunchKernelGGL(ks[v], dim3(1), dim3(NT), 0, 0, din, o0);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(r0.data(), o0, 4 * NT * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < 4 * NT; ++i) if (r0[i] != r1[i]) ++badv[v];
        }
        tot += 4 * NT;
    }
    const char* nm[7] = {"none", "s_nop 0", "s_nop 1", "s_nop 3", "full v_mov then MFMA", "v_mov + s_nop 0", "v_mov + s_nop 1"};
    for (int v = 0; v < 7; ++v) printf("%-24s: %ld of %ld differ\n", nm[v], badv[v], tot);
    return 0;
}
