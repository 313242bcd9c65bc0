// Harness for the split-fp16 main loop (tile_f16.h): host-side split, fp64 check, timing.
//   /tmp/bench_f16 B N d RS iters
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "tile_f16.h"
using namespace nw;

template <int RS>
__global__ __launch_bounds__(TILE_THREADS, (RS <= 5 ? 4 : 2)) void dots_kernel(const float* q, const float* s, float* out,
                                                                              int B, int N, int d, int n_stiles, int n_qtiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* stage = reinterpret_cast<float4*>(smem);
    int qt, st;
    if (!decode_block(n_stiles, n_qtiles, qt, st)) return;
    f32x4 acc[RS];
    tile_dots_f16x2<RS>(q, s, B, N, d, qt * BQ, st * 16 * RS, stage, acc, st % (d / 32));
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
    if (wave >= NCONS) return;
    const int b = qt * BQ + 16 * wave + i;
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = st * 16 * RS + 16 * r + 4 * g + e;
#ifdef NW_BENCH_NOSTORE
            if (b < B && j < N && acc[r][e] == 12345.678f) out[(size_t)b * N + j] = acc[r][e];
#else
            if (b < B && j < N) out[(size_t)b * N + j] = acc[r][e];
#endif
        }
}

// host split: rows scaled by 2^e (rowmax -> [2^13, 2^14)), every 32-float chunk -> [32 h | 32 l] fp16
static void split_rows(const std::vector<float>& x, int rows, int d, std::vector<float>& out, std::vector<float>& scale) {
    out.resize(x.size()); scale.resize(rows);
    for (int r = 0; r < rows; ++r) {
        float mx = 0; for (int k = 0; k < d; ++k) mx = fmaxf(mx, fabsf(x[(size_t)r * d + k]));
        int e = 0; if (mx > 0) { frexpf(mx, &e); e = 14 - e; }   // mx * 2^e in [2^13, 2^14)
        scale[r] = ldexpf(1.f, -e);
        for (int c = 0; c < d / 32; ++c) {
            _Float16* dst = reinterpret_cast<_Float16*>(&out[(size_t)r * d + c * 32]);
            for (int k = 0; k < 32; ++k) {
                const float v = ldexpf(x[(size_t)r * d + c * 32 + k], e);
                const _Float16 h = (_Float16)v;
                const _Float16 l = (_Float16)(v - (float)h);
                dst[k] = h; dst[32 + k] = l;
            }
        }
    }
}

template <int RS>
float run(const float* q, const float* s, float* out, int B, int N, int d, int iters) {
    const int n_stiles = (N + 16 * RS - 1) / (16 * RS), n_qtiles = (B + BQ - 1) / BQ;
    const int grid = padded_grid(n_stiles, n_qtiles);
    const size_t lds = DmaCfg<RS>::STAGE_BYTES;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL(dots_kernel<RS>, dim3(grid), dim3(TILE_THREADS), lds, 0, q, s, out, B, N, d, n_stiles, n_qtiles); };
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < iters; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("grid=%d lds=%zu ", grid, lds);
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    int B = atoi(argv[1]), N = atoi(argv[2]), d = atoi(argv[3]), RS = atoi(argv[4]), iters = atoi(argv[5]);
    const char* dist = argc > 6 ? argv[6] : "uniform";
    std::vector<float> hq((size_t)B * d), hs((size_t)N * d);
    srand(1);
    auto rnd = [&] { return (rand() / (float)RAND_MAX) * 2 - 1; };
    for (auto& v : hq) v = rnd();
    for (auto& v : hs) v = rnd();
    if (!strcmp(dist, "wide")) {   // wild dynamic range: rows at 1e-6 .. 1e6, elements spanning 1e5 inside a row
        for (int r = 0; r < B; ++r) { float sc = powf(10.f, (r % 13) - 6.f); for (int k = 0; k < d; ++k) hq[(size_t)r * d + k] *= sc * powf(10.f, -(k % 6)); }
        for (int r = 0; r < N; ++r) { float sc = powf(10.f, (r % 11) - 5.f); for (int k = 0; k < d; ++k) hs[(size_t)r * d + k] *= sc * powf(10.f, -(k % 5)); }
    }
    std::vector<float> sq, ss, scq, scs;
    split_rows(hq, B, d, sq, scq); split_rows(hs, N, d, ss, scs);
    float *q, *s, *out;
    hipMalloc(&q, sq.size() * 4); hipMalloc(&s, ss.size() * 4); hipMalloc(&out, (size_t)B * N * 4);
    hipMemcpy(q, sq.data(), sq.size() * 4, hipMemcpyHostToDevice); hipMemcpy(s, ss.data(), ss.size() * 4, hipMemcpyHostToDevice);
    float us = 0;
    switch (RS) {
        case 5: us = run<5>(q, s, out, B, N, d, iters); break;
        case 8: us = run<8>(q, s, out, B, N, d, iters); break;
        case 10: us = run<10>(q, s, out, B, N, d, iters); break;
        default: us = run<12>(q, s, out, B, N, d, iters); break;
    }
    printf("B=%d N=%d d=%d RS=%d : %.2f us  %.1f TFLOP/s-equivalent (%.0f%% of the fp32-MFMA peak 157.3)\n", B, N, d, RS, us,
           2.0 * B * N * d / us / 1e6, 2.0 * B * N * d / us / 1e6 / 157.3 * 100);
    // accuracy on a sample of pairs vs fp64, relative to sum |a_k b_k|
    std::vector<float> ho((size_t)B * N);
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst32 = 0;
    for (int t = 0; t < 4000; ++t) {
        const int b = rand() % B, j = rand() % N;
        double ref = 0, mag = 0; float f32 = 0;
        for (int k = 0; k < d; ++k) { const double p = (double)hq[(size_t)b * d + k] * hs[(size_t)j * d + k]; ref += p; mag += fabs(p); f32 = fmaf(hq[(size_t)b * d + k], hs[(size_t)j * d + k], f32); }
        const double got = (double)ho[(size_t)b * N + j] * scq[b] * scs[j];
        worst = fmax(worst, fabs(got - ref) / mag); worst32 = fmax(worst32, fabs((double)f32 - ref) / mag);
    }
    printf("   max |dot - fp64| / sum|a_k b_k| over 4000 pairs: split-fp16 %.3e   (plain fp32 FMA chain %.3e)\n", worst, worst32);
    return 0;
}
