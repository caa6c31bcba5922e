// Probe of v_mfma_f32_4x4x1_16b_f32 operand layout and of the CBSZ / ABID broadcast (used by csrc/hjbq_kernels.h).
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4_probe tools/mfma4_probe.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ABID>
__global__ void probe(const float* a, const float* b, float* out) {
    const int lane = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], c, 4, ABID, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}
__global__ void probe_plain(const float* a, const float* b, float* out) {
    const int lane = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}
int main() {
    float ha[64], hb[64], ho[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 100.f + l; hb[l] = 1.f + 0.001f * l; }   // a: 100 + lane, b: 1 + lane / 1000
    float *a, *b, *o;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    // expectation under test: out[lane = (blk, c)][i] = a[(blk_a, i)] * b[(blk, c)], blk_a = blk (plain) or ABID (cbsz = 4)
    int bad = 0;
    probe_plain<<<1, 64>>>(a, b, o); hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const float want = ha[(l / 4) * 4 + i] * hb[l];
        if (ho[l * 4 + i] != want) { if (bad < 8) printf("plain lane %d reg %d: got %g want %g\n", l, i, ho[l * 4 + i], want); ++bad; }
    }
    printf("plain: %d mismatches\n", bad);
    int bad5 = 0;
    probe<5><<<1, 64>>>(a, b, o); hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const float want = ha[5 * 4 + i] * hb[l];
        if (ho[l * 4 + i] != want) { if (bad5 < 8) printf("abid5 lane %d reg %d: got %g want %g\n", l, i, ho[l * 4 + i], want); ++bad5; }
    }
    printf("cbsz=4 abid=5: %d mismatches\n", bad5);
    int bad15 = 0;
    probe<15><<<1, 64>>>(a, b, o); hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const float want = ha[15 * 4 + i] * hb[l];
        if (ho[l * 4 + i] != want) ++bad15;
    }
    printf("cbsz=4 abid=15: %d mismatches\n", bad15);
    return (bad || bad5 || bad15) ? 1 : 0;
}
