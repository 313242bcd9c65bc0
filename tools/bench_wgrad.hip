// Timing / ablation harness of nw_conv_wgrad_kernel (build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DNW_WABL_NOMFMA |
// NOLOAD | NOCVT | NOFRAG] -Inwhead_amd/csrc -Iinclude -o tools/bench_wgrad tools/bench_wgrad.hip).  usage: bench_wgrad n cin h w cout k
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nw_internal.h"
namespace nw { int knob(int) { return KNOB_UNSET; } }
static float4 g_zero_host;
__device__ float4 g_zero_dev[2];
extern "C" const void* nw_conv_zero_page(void) {
    void* ptr = nullptr;
    return hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_zero_dev)) == hipSuccess ? ptr : nullptr;
}
#include "conv_wgrad.hip"
int main(int argc, char** argv) {
    if (argc < 7) { std::printf("usage: bench_wgrad n cin h w cout k\n"); return 1; }
    const int64_t n = atoi(argv[1]), cin = atoi(argv[2]), h = atoi(argv[3]), w = atoi(argv[4]), cout = atoi(argv[5]), k = atoi(argv[6]);
    const int64_t pad = k / 2;
    std::vector<float> hx((size_t)n * h * w * cin), hg((size_t)n * h * w * cout), ham(256, 1.0f);
    srand(1);
    for (auto& v : hx) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hg) v = (rand() / (float)RAND_MAX) * 2 - 1;
    float *x, *g, *dw, *am; void* ws;
    const size_t wsb = nw_conv2d_nhwc_wgrad_workspace_bytes(n, h, w, cin, cout, k, k, 1, pad);
    hipMalloc(&x, hx.size() * 4); hipMalloc(&g, hg.size() * 4); hipMalloc(&dw, (size_t)cout * k * k * cin * 4); hipMalloc(&am, 1024);
    hipMalloc(&ws, wsb ? wsb : 16);
    hipMemset(g_zero_dev, 0, 0);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(am, ham.data(), 1024, hipMemcpyHostToDevice);
    auto launch = [&] { return nw_conv2d_nhwc_wgrad_f16x2(x, am, g, am, dw, ws, wsb, n, h, w, cin, cout, k, k, 1, pad, 0, 0, nullptr); };
    int rc = launch();
    if (rc) { std::printf("launch failed: %d\n", rc); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::printf("wgrad + reduce %.2f us per call\n", ms * 1e3 / 50);
    return 0;
}
