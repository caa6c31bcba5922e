// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the path-store kernels use
// (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of 16 B/lane streams; other widths are uncalibrated).
// Each kernel streams a buffer of known size exactly once:
//   read4   one dword per lane, lane-linear (256 B per wave instruction)   -- T-layout image reads
//   read16  16 B per lane (1 KiB per wave instruction)                     -- feature-on-lane image reads
//   read8   8 B per lane
//   write4 / write16                                                        -- path-store writes
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./tools/fetch_calibrate   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void read4(const float* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456f) out[0] = s;
}
__global__ void read8(const float2* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float2 v = p[i]; s += v.x + v.y; }
    if (s == 123.456f) out[0] = s;
}
__global__ void read16(const float4* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}
__global__ void write4(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f;
}
__global__ void write16(float4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const size_t bytes = (size_t)2 << 30;             // 2 GiB: far beyond the 256 MiB Infinity Cache
    float *buf, *out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes));
    const int grid = 256 * 8, block = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read4, dim3(grid), dim3(block), 0, 0, buf, bytes / 4, out);
        hipLaunchKernelGGL(read8, dim3(grid), dim3(block), 0, 0, (const float2*)buf, bytes / 8, out);
        hipLaunchKernelGGL(read16, dim3(grid), dim3(block), 0, 0, (const float4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(write4, dim3(grid), dim3(block), 0, 0, buf, bytes / 4);
        hipLaunchKernelGGL(write16, dim3(grid), dim3(block), 0, 0, (float4*)buf, bytes / 16);
    }
    CK(hipDeviceSynchronize());
    printf("each kernel streams %zu bytes once\n", bytes);
    return 0;
}
