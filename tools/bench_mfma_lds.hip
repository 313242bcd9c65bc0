// What does an LDS-fed fp32 MFMA loop cost on gfx950?  Variants (one wave per SIMD, 4 waves/WG):
//   0: MFMA only, operands in 11 distinct registers
//   1: + 11 ds_read_b128 per 40 MFMAs, results unused
//   2: MFMA operands = results of the ds_reads issued one step earlier (software pipelined)
//   3: like 2 but all four waves read the SAME A fragments (as the real kernel does)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RS = 10;

template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float4 lds[4096];   // 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int x = tid; x < 4096; x += 256) lds[x] = make_float4(x * 1e-3f, 1.f, -x * 1e-3f, 0.5f);
    __syncthreads();
    f32x4 acc[RS];
    for (int r = 0; r < RS; ++r) acc[r] = f32x4{0, 0, 0, 0};
    float4 a[RS], b, a2[RS], b2;
    const int i = lane & 15, g = lane >> 4;
    const int slot = g ^ ((i >> 1) & 7);
    const int wbase = (VAR == 3) ? 0 : wave * 1024;
    const float4* base = lds + wbase + i * 8 + slot;
    for (int r = 0; r < RS; ++r) a[r] = base[r * 128];
    b = lds[wave * 128 + 2048 + i * 8 + slot];
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 1) {
            const float4* p = base + ((it & 3) * 4 % 8);
#pragma unroll
            for (int r = 0; r < RS; ++r) a2[r] = p[r * 128];
            b2 = lds[wave * 128 + 2048 + i * 8 + (slot ^ (it & 1))];
        }
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].x, b.x, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].y, b.y, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].z, b.z, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].w, b.w, acc[r], 0, 0, 0);
        if (VAR == 1) {
#pragma unroll
            for (int r = 0; r < RS; ++r) asm volatile("" ::"v"(a2[r].x), "v"(a2[r].y), "v"(a2[r].z), "v"(a2[r].w));
            asm volatile("" ::"v"(b2.x), "v"(b2.y), "v"(b2.z), "v"(b2.w));
        }
        if (VAR >= 2) {
#pragma unroll
            for (int r = 0; r < RS; ++r) a[r] = a2[r];
            b = b2;
        }
    }
    float t = 0;
    for (int r = 0; r < RS; ++r) t += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[blockIdx.x * 256 + tid] = t;
}
template <int VAR>
void run(float* out, int wg, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(wg), dim3(256), 0, 0, out, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int x = 0; x < 10; ++x) hipLaunchKernelGGL(k<VAR>, dim3(wg), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double flop = (double)wg * 4 * iters * 40 * 2048;
    printf("var %d grid %4d: %8.2f us  %6.1f TFLOP/s  (%.1f cycles/MFMA at 2.3 GHz)\n", VAR, wg, ms * 1e3, flop / ms / 1e9,
           ms * 1e-3 * 2.3e9 / (iters * 40.0) / ((wg + 255) / 256));
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 24);
    for (int wg : {256, 512}) { run<0>(out, wg, 2000); run<1>(out, wg, 2000); run<2>(out, wg, 2000); run<3>(out, wg, 2000); }
    return 0;
}
