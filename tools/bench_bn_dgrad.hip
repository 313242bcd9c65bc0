// Harness for csrc/bn_dgrad.hip: time of nw_bn_dgrad1x1_bwd_f16x2 (pass 1 + finalize + pass 2) per shape.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Inwhead_amd/csrc -Iinclude -o tools/bench_bn_dgrad tools/bench_bn_dgrad.hip
// usage: bench_bn_dgrad rows c [k=128] [ctot=c]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nw_internal.h"
namespace nw { int knob(int) { return KNOB_UNSET; } }
#include "bn_nhwc.hip"
#include "bn_dgrad.hip"
int main(int argc, char** argv) {
    if (argc < 3) { std::printf("usage: bench_bn_dgrad rows c [k] [ctot]\n"); return 1; }
    const int64_t rows = atoll(argv[1]), c = atoll(argv[2]), k = argc > 3 ? atoll(argv[3]) : 128, ctot = argc > 4 ? atoll(argv[4]) : c;
    std::vector<float> h((size_t)rows * ctot);
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2 - 1;
    float *du, *amax, *ws, *wsc, *x, *tab, *inv, *G, *amo, *dg, *db, *wk;
    hipMalloc(&du, rows * k * 4); hipMalloc(&amax, 1024); hipMalloc(&ws, c * k * 4); hipMalloc(&wsc, c * 4);
    hipMalloc(&x, rows * ctot * 4); hipMalloc(&tab, 3 * c * 4); hipMalloc(&inv, c * 4); hipMalloc(&G, rows * ctot * 4);
    hipMalloc(&amo, 1024); hipMalloc(&dg, c * 4); hipMalloc(&db, c * 4);
    const size_t wb = nw_bn_dgrad1x1_workspace_bytes(rows, c);
    hipMalloc(&wk, wb);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(G, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(du, h.data(), std::min(h.size(), (size_t)(rows * k)) * 4, hipMemcpyHostToDevice);
    std::vector<float> one(std::max<int64_t>(3 * c, 256), 1.f);
    hipMemcpy(amax, one.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(wsc, one.data(), c * 4, hipMemcpyHostToDevice);
    hipMemcpy(inv, one.data(), c * 4, hipMemcpyHostToDevice);
    for (int64_t i = 0; i < c; ++i) one[i] = 0.f;                 // mean 0, a 1, beta 1
    hipMemcpy(tab, one.data(), 3 * c * 4, hipMemcpyHostToDevice);
    hipMemset(ws, 0, c * k * 4);
    auto launch = [&] { return nw_bn_dgrad1x1_bwd_f16x2(du, amax, ws, wsc, x, ctot, tab, c, inv, G, ctot, amo, dg, db, wk, wb, rows, c, k, nullptr); };
    int rc = launch();
    if (rc) { std::printf("failed: %d\n", rc); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 30; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 30, bytes = (double)rows * (4.0 * c + 2.0 * k) * 4;
    std::printf("rows %lld c %lld k %lld: %.1f us per call (both passes + finalize), %.0f GB/s of algorithmic traffic\n", (long long)rows, (long long)c,
                (long long)k, us, bytes / us * 1e-3);
    return 0;
}
