// Pure-register fp32 MFMA rate on gfx950: NACC independent accumulators per wave, WPS waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int r = 0; r < NACC; ++r) acc[r] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[r], 0, 0, 0);
    }
    float t = 0;
    for (int r = 0; r < NACC; ++r) t += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int r = 0; r < NACC; ++r) for (int e = 0; e < 16; ++e) acc[r][e] = 0;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
    }
    float t = 0;
    for (int r = 0; r < NACC; ++r) for (int e = 0; e < 16; ++e) t += acc[r][e];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <typename F>
void timeit(const char* name, F launch, double flop) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-28s %8.2f us  %7.1f TFLOP/s\n", name, ms * 1e3, flop / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    const int iters = 1000;
    for (int wg = 256; wg <= 512; wg *= 2) {
        printf("grid=%d (x256 threads)\n", wg);
        timeit("16x16x4  NACC=2", [&] { hipLaunchKernelGGL(k16<2>, dim3(wg), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, (double)wg * 4 * iters * 4 * 2 * 2048);
        timeit("16x16x4  NACC=4", [&] { hipLaunchKernelGGL(k16<4>, dim3(wg), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, (double)wg * 4 * iters * 4 * 4 * 2048);
        timeit("16x16x4  NACC=10", [&] { hipLaunchKernelGGL(k16<10>, dim3(wg), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, (double)wg * 4 * iters * 4 * 10 * 2048);
        timeit("32x32x2  NACC=1", [&] { hipLaunchKernelGGL(k32<1>, dim3(wg), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, (double)wg * 4 * iters * 4 * 1 * 4096);
        timeit("32x32x2  NACC=4", [&] { hipLaunchKernelGGL(k32<4>, dim3(wg), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, (double)wg * 4 * iters * 4 * 4 * 4096);
    }
    return 0;
}
