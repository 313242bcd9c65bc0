// Standalone micro-benchmark of the shared main loop (tile_core.h): no torch, hipEvent timing.
//   hipcc -O3 --offload-arch=gfx950 -I nwhead_amd/csrc [-DNW_ABL_...] tools/bench_tile.hip -o /tmp/bench_tile
//   /tmp/bench_tile B N d RS iters
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include "tile_core.h"
#include "tile_dma.h"

using namespace nw;
#ifndef NW_NORM
#define NW_NORM true
#endif
#ifndef NW_SNORM
#define NW_SNORM true
#endif
#ifndef NW_ROT
#define NW_ROT 0
#endif

template <int RS>
__global__ __launch_bounds__(TILE_THREADS, (RS <= 5 ? 4 : 2)) void dots_kernel(const float* q, const float* s, float* out, int B, int N, int d,
                                                    int n_stiles, int n_qtiles, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qn2 = reinterpret_cast<float*>(smem);
    float* sn2 = qn2 + 64;
    float4* stage = reinterpret_cast<float4*>(smem + 1024);
    int qt, st;
    if (!decode_block(n_stiles, n_qtiles, qt, st)) return;
    f32x4 acc[RS];
#ifdef NW_DIAG_CLOCK
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef NW_USE_DMA
    tile_dots_dma<RS, NW_NORM, NW_SNORM>(q, s, B, N, d, qt * BQ, st * 16 * RS, stage, qn2, sn2, acc, NW_ROT);
#else
    tile_dots<RS, NW_NORM>(q, s, B, N, d, qt * BQ, st * 16 * RS, stage, qn2, sn2, acc, NW_ROT, clk + 8192);
#endif
#ifdef NW_DIAG_CLOCK
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
#endif
    float t = qn2[threadIdx.x & 63] + sn2[threadIdx.x % (16 * RS)];
#pragma unroll
    for (int r = 0; r < RS; ++r) t += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[(size_t)blockIdx.x * TILE_THREADS + threadIdx.x] = t;
}

static unsigned long long* g_clk = nullptr;
template <int RS>
float run(const float* q, const float* s, float* out, int B, int N, int d, int iters) {
    const int n_stiles = (N + 16 * RS - 1) / (16 * RS), n_qtiles = (B + BQ - 1) / BQ;
    const int grid = padded_grid(n_stiles, n_qtiles);
    #ifdef NW_USE_DMA
    const size_t lds = 1024 + DmaCfg<RS>::STAGE_BYTES;
#else
    const size_t lds = 1024 + TileCfg<RS>::STAGE_BYTES;
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(dots_kernel<RS>, dim3(grid), dim3(TILE_THREADS), lds, 0, q, s, out, B, N, d, n_stiles, n_qtiles, g_clk);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(dots_kernel<RS>, dim3(grid), dim3(TILE_THREADS), lds, 0, q, s, out, B, N, d, n_stiles, n_qtiles, g_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("grid=%d lds=%zu ", grid, lds);
#ifdef NW_DIAG_CLOCK
    {
        std::vector<unsigned long long> h(2 * grid);
        hipMemcpy(h.data(), g_clk, h.size() * 8, hipMemcpyDeviceToHost);
        double sc = 0, sr = 0; int n = 0;
        for (int i = 0; i < grid; ++i) if (h[2 * i + 1]) { sc += h[2 * i]; sr += h[2 * i + 1]; ++n; }
        printf("[in-kernel clock %.3f GHz, main loop %.0f cycles avg over %d WGs] ", sc / sr * 0.1, sc / n, n);
#ifdef NW_DIAG_PHASES
        std::vector<unsigned long long> p(4 * grid);
        hipMemcpy(p.data(), g_clk + 8192, p.size() * 8, hipMemcpyDeviceToHost);
        double a[4] = {0, 0, 0, 0};
        for (int i = 0; i < grid; ++i) for (int k = 0; k < 4; ++k) a[k] += p[4 * i + k];
        printf("\n   loader: wait+store %.0f, issue loads %.0f, barrier wait %.0f | consumer barrier wait %.0f (cycles per WG, sum over stages)\n   ",
               a[0] / n, a[1] / n, a[2] / n, a[3] / n);
#endif
    }
#endif
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    int B = atoi(argv[1]), N = atoi(argv[2]), d = atoi(argv[3]), RS = atoi(argv[4]), iters = atoi(argv[5]);
    std::vector<float> hq((size_t)B * d), hs((size_t)N * d);
    srand(1);
    for (auto& v : hq) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hs) v = (rand() / (float)RAND_MAX) * 2 - 1;
    float *q, *s, *out;
    hipMalloc(&q, hq.size() * 4); hipMalloc(&s, hs.size() * 4); hipMalloc(&out, (size_t)64 << 20); hipMalloc(&g_clk, 1 << 20); hipMemset(g_clk, 0, 1 << 20);
    hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    float us = 0;
    switch (RS) {
        case 4: us = run<4>(q, s, out, B, N, d, iters); break;
        case 6: us = run<6>(q, s, out, B, N, d, iters); break;
        case 5: us = run<5>(q, s, out, B, N, d, iters); break;
        case 3: us = run<3>(q, s, out, B, N, d, iters); break;
        case 8: us = run<8>(q, s, out, B, N, d, iters); break;
        case 10: us = run<10>(q, s, out, B, N, d, iters); break;
        default: us = run<12>(q, s, out, B, N, d, iters); break;
    }
    printf("B=%d N=%d d=%d RS=%d : %.2f us  %.1f TFLOP/s (%.1f%% of 157.3)\n", B, N, d, RS, us,
           2.0 * B * N * d / us / 1e6, 2.0 * B * N * d / us / 1e6 / 157.3 * 100);
    return 0;
}
