// Diagnostic harness for nw_xgemm_kernel (bwd_split.hip): the two products of the backward at B N d, timing only.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -Inwhead_amd/csrc [-DXG_NO_MFMA|-DXG_NO_DMA] tools/bench_xgemm.hip nwhead_amd/csrc/split.hip -o build/bench_xgemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "bwd_split.hip"
using namespace nw;
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 10000, d = argc > 3 ? atoi(argv[3]) : 512;
    const int ld = (N + 31) / 32 * 32, Bpad = (B + 31) / 32 * 32;
    std::vector<float> hA((size_t)B * ld), hs((size_t)N * d), hq((size_t)Bpad * d, 0.f);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hs) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (size_t i = 0; i < (size_t)B * d; ++i) hq[i] = (rand() / (float)RAND_MAX) * 2 - 1;
    float *A, *As, *s, *ss, *q, *qs, *sc, *n2, *part, *gq, *gs, *one;
    hipMalloc(&A, hA.size() * 4 + 512); hipMalloc(&As, hA.size() * 4 + 512);
    hipMalloc(&s, hs.size() * 4 + 512); hipMalloc(&ss, hs.size() * 4 + 512);
    hipMalloc(&q, hq.size() * 4 + 512); hipMalloc(&qs, hq.size() * 4 + 512);
    hipMalloc(&sc, (size_t)(N + ld) * 4); hipMalloc(&n2, (size_t)(N + ld) * 4);
    hipMalloc(&part, (size_t)64 * B * d * 4 + (size_t)4 * N * d * 4); hipMalloc(&gq, (size_t)B * d * 4); hipMalloc(&gs, (size_t)N * d * 4);
    hipMalloc(&one, 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    const float f1 = 1.f; hipMemcpy(one, &f1, 4, hipMemcpyHostToDevice);
    launch_split_rows(A, As, sc, n2, B, ld, 0);
    launch_split_rows(q, qs, sc, n2, Bpad, d, 0);
    launch_split_rows(s, ss, sc, n2, N, d, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        auto launch = [&] {
            if (which == 0) launch_xgemm(false, As, ld, B, ss, d, N, part, sc, 0, nullptr, n2, q, gq, B, d, N, 0);
            else launch_xgemm(true, As, ld, B, qs, d, Bpad, part, sc, 1, one, n2, s, gs, N, d, B, 0);
        };
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 100; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const XgemmPlan p = which == 0 ? xgemm_plan(B, d, N) : xgemm_plan(N, d, B);
#ifdef XG_DIAG
        {
            std::vector<unsigned long long> h(8 * 4096);
            hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(nw_diag_x), h.size() * 8);
            double ph[5] = {0}; int n = 0; unsigned long long tmin = ~0ull, tmax = 0;
            for (int b = 0; b < 4096; ++b) {
                const unsigned long long* t = &h[8 * b];
                if (!t[0] || !t[4]) continue;
                ++n;
                ph[0] += (double)(t[1] - t[0]); ph[1] += (double)(t[2] - t[1]); ph[2] += (double)(t[3] - t[2]);
                ph[3] += (double)(t[4] - t[3]); ph[4] += (double)(t[5] - t[0]);
                tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[4]);
            }
            printf("   ticks per workgroup (n=%d): fill %.0f, loop %.0f, staging %.0f, finish %.0f | loaders done with DMA at %.0f | first start -> last end %.0f\n",
                   n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, (double)(tmax - tmin));
            hipMemset(nullptr, 0, 0);
            std::vector<unsigned long long> z(8 * 4096, 0);
            hipMemcpyToSymbol(HIP_SYMBOL(nw_diag_x), z.data(), z.size() * 8);
        }
#endif
        printf("%s: %.2f us per call (kernel + reduce if split; %d chunks of %d)  %s\n", which == 0 ? "gq = A' s' " : "gs = A'^T q''",
               ms * 10, p.nchunks, p.k_chunk, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
