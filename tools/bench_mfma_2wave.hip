// Two MFMA-issuing waves per SIMD vs one: cycles per v_mfma_f32_16x16x32_f16 seen by a wave (s_memtime), with and without
// an s_barrier every BLK MFMAs.  usage: bench_mfma_2wave   (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int NACC, int BLK, bool BAR>
__global__ __launch_bounds__(512, 2) void kern(unsigned long long* out, float* sink, int iters, float seed) {
    f32x4 acc[NACC];
    for (int r = 0; r < NACC; ++r) acc[r] = f32x4{0, 0, 0, 0};
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + threadIdx.x * 0.001f + e); b[e] = (_Float16)(seed - threadIdx.x * 0.002f - e); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < BLK / NACC; ++c)
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float t = 0;
    for (int r = 0; r < NACC; ++r) t += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int BLK, bool BAR>
void run(const char* name, int threads) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
    hipMemset(out, 0, 256 * 8 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((kern<NACC, BLK, BAR>), dim3(256), dim3(threads), 0, 0, out, sink, iters, 1.0f);
    hipLaunchKernelGGL((kern<NACC, BLK, BAR>), dim3(256), dim3(threads), 0, 0, out, sink, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { s += (double)h[b * 8 + w]; ++n; }
    const double per_wave = s / n / ((double)iters * BLK);
    const int wps = threads / 256;
    printf("%-44s waves/SIMD %d  cycles per MFMA per wave %6.2f  -> per SIMD %6.2f\n", name, wps, per_wave, per_wave / wps);
    hipFree(out); hipFree(sink);
}
int main() {
    run<16, 48, false>("16 acc, no barrier", 256);
    run<16, 48, false>("16 acc, no barrier", 512);
    run<16, 48, true>("16 acc, barrier per 48", 256);
    run<16, 48, true>("16 acc, barrier per 48", 512);
    run<16, 96, true>("16 acc, barrier per 96", 256);
    run<16, 96, true>("16 acc, barrier per 96", 512);
    run<4, 48, false>("4 acc, no barrier", 256);
    run<4, 48, false>("4 acc, no barrier", 512);
    run<2, 48, false>("2 acc, no barrier", 256);
    run<2, 48, false>("2 acc, no barrier", 512);
    run<1, 48, false>("1 acc, no barrier", 256);
    run<1, 48, false>("1 acc, no barrier", 512);
    return 0;
}
