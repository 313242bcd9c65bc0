// Diagnostic harness: phase stamps of nw_conv_nhwc_kernel (build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DNW_CONV_DIAG
// -Inwhead_amd/csrc -Iinclude -o tools/bench_conv tools/bench_conv.hip).  usage: bench_conv n cin h w cout k [pad] [ldx ldy]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "nw_internal.h"
namespace nw {   // the library's knobs from the environment here: NW_CONV_SKIP_CFGS=n passes over the first n fitting tile shapes
int knob(int id) {
    const char* e = id == KNOB_CONV_SKIP_CFGS ? getenv("NW_CONV_SKIP_CFGS") : id == KNOB_CONV_FORCE_CFG ? getenv("NW_CONV_FORCE_CFG") : nullptr;
    return e ? atoi(e) : KNOB_UNSET;
}
}
#include "conv_nhwc.hip"
int main(int argc, char** argv) {
    if (argc < 7) { std::printf("usage: bench_conv n cin h w cout k [pad]\n"); return 1; }
    const int64_t n = atoi(argv[1]), cin = atoi(argv[2]), h = atoi(argv[3]), w = atoi(argv[4]), cout = atoi(argv[5]), k = atoi(argv[6]);
    const int64_t pad = argc > 7 ? atoi(argv[7]) : k / 2;
    const int64_t ho = h + 2 * pad - k + 1, wo = w + 2 * pad - k + 1;
    std::vector<float> hx((size_t)n * h * w * cin);
    srand(1);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
    // split weight rows: per 128-byte chunk [32 h | 32 l] fp16; h = small random, l = 0 (timing only)
    const size_t wcols = (size_t)k * k * cin;
    std::vector<__half> hw(cout * wcols * 2);
    for (size_t r = 0; r < (size_t)cout; ++r)
        for (size_t c = 0; c < wcols; c += 32)
            for (int j = 0; j < 32; ++j) {
                hw[(r * wcols + c) * 2 + j] = __float2half(((rand() / (float)RAND_MAX) * 2 - 1) * 8192.f);
                hw[(r * wcols + c) * 2 + 32 + j] = __float2half((rand() / (float)RAND_MAX) * 4.f);
            }
    std::vector<float> hs(cout, 1.f / 8192.f), ham(256, 1.0f);
    float *x, *y, *ws, *sc, *am, *amo;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&y, (size_t)n * ho * wo * cout * 4); hipMalloc(&ws, hw.size() * 2); hipMalloc(&sc, cout * 4);
    hipMalloc(&am, 1024); hipMalloc(&amo, 1024);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ws, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(sc, hs.data(), cout * 4, hipMemcpyHostToDevice); hipMemcpy(am, ham.data(), 1024, hipMemcpyHostToDevice);
    // NW_BC_MOMENTS=1: with the moments epilogue; NW_BC_PRE=1: BatchNorm + ReLU in the loaders (identity table)
    float* mom = nullptr; float* tab = nullptr;
    if (getenv("NW_BC_MOMENTS")) hipMalloc(&mom, (size_t)5 * 4 * nw_conv2d_nhwc_moments_groups(n, h, w, cin, cout, k, k, 1, pad) * cout + 64);
    if (getenv("NW_BC_PRE")) {
        std::vector<float> ht(3 * cin, 0.f);
        for (int64_t c = 0; c < cin; ++c) ht[cin + c] = 1.f;
        hipMalloc(&tab, ht.size() * 4); hipMemcpy(tab, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    }
    auto launch = [&] {
        if (tab) return nw_conv2d_nhwc_bnrelu_f16x2(x, tab, am, 0, ws, sc, nullptr, 0, y, amo, n, h, w, cin, cout, k, k, 1, pad, 0, 0, mom, nullptr);
        return nw_conv2d_nhwc_f16x2(x, am, ws, sc, nullptr, nullptr, 0, y, amo, n, h, w, cin, cout, k, k, 1, pad, 0, 0, mom, nullptr);
    };
    int rc = launch();
    if (rc) { std::printf("launch failed: %d\n", rc); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
#ifndef NW_CONV_DIAG   // plain or ablation build (-DNW_CABL_NOMFMA / NOLOAD / NOCVT): the time only
    std::printf("kernel %.2f us\n", ms * 1e3 / 50);
    return 0;
#else
    std::vector<unsigned long long> d(16 * 1024);
    hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(nw::nw_conv_diag), d.size() * 8);
    double L[7] = {0}, Cn[6] = {0}; int nw_ = 0;
    for (int b = 0; b < 1024; ++b) {
        if (!d[16 * b + 6]) continue;
        ++nw_;
        for (int k2 = 0; k2 < 7; ++k2) L[k2] += (double)d[16 * b + k2];
        for (int k2 = 0; k2 < 5; ++k2) Cn[k2] += (double)d[16 * b + 8 + k2];
        Cn[5] += (double)d[16 * b + 14];
    }
    std::printf("kernel %.2f us (diag build: stamps inflate it) | %d workgroups, ticks per workgroup:\n", ms * 1e3 / 50, nw_);
    std::printf("  loader  : bookkeeping+issue_w %.0f | wait chunk loads %.0f | convert+LDS stores %.0f | issue chunk loads %.0f | wait weights %.0f | barrier %.0f | total %.0f\n",
                L[0] / nw_, L[1] / nw_, L[2] / nw_, L[3] / nw_, L[4] / nw_, L[5] / nw_, L[6] / nw_);
    std::printf("  consumer: fill wait %.0f | MFMA+LDS reads %.0f | barrier %.0f | tile set-up %.0f | epilogue %.0f | total %.0f\n",
                Cn[0] / nw_, Cn[1] / nw_, Cn[2] / nw_, Cn[3] / nw_, Cn[4] / nw_, Cn[5] / nw_);
    return 0;
#endif
}
