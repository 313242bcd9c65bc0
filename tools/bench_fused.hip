// Diagnostic harness: phase stamps of nw_fused_kernel (build with -DNW_DIAG_FUSED).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "fused_impl.h"
using namespace nw;
namespace nw {
size_t fused_layout(int64_t, int64_t, int, char*, FusedWs*, int64_t) { return 0; }
int launch_merge_runs(const FusedWs&, float*, float*, float*, float*, float*, int, int, int, int, hipStream_t) { return 0; }
int launch_run_tables(const FusedWs&, const int64_t*, int, int, int, int, hipStream_t) { return 0; }  // tables are built on the host below
int pick_rs(int64_t, int64_t, int64_t, bool) { return 10; }
int device_cu_count() { return 256; }
bool env_flag(const char*) { return false; }
int persistent_variant() { const char* e = getenv("NW_PVAR"); return e ? atoi(e) : 2; }
int persistent_qgroup() { const char* e = getenv("NW_QG"); return e ? atoi(e) : 8; }
const FwdOpts& fwd_opts() { static FwdOpts o{}; return o; }
int knob(int) { return KNOB_UNSET; }
int launch_split_rows(const float* x, float* out, float* scale, float* norm2, int64_t rows, int64_t d, hipStream_t st);
}
int main(int argc, char** argv) {
    const int B = atoi(argv[1]), N = atoi(argv[2]), d = atoi(argv[3]), C = atoi(argv[4]);
    #ifndef NW_BENCH_RS
#define NW_BENCH_RS 10
#endif
    constexpr int RS = NW_BENCH_RS, BS = 16 * RS;
    std::vector<float> hq((size_t)B * d), hs((size_t)N * d);
    srand(1);
    for (auto& v : hq) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hs) v = (rand() / (float)RAND_MAX) * 2 - 1;
    std::vector<int64_t> hy(N);
    for (int j = 0; j < N; ++j) hy[j] = (int64_t)j * C / N;
    const int n_stiles = (N + BS - 1) / BS, n_qtiles = (B + 63) / 64;
    const bool f16 = argc > 5 && atoi(argv[5]) >= 1;
    const bool persistent = argc > 5 && atoi(argv[5]) == 2;
    const bool qraw = argc > 5 && atoi(argv[5]) == 3;   // MODE_F16Q: raw fp32 queries, split in the consumer waves
    float *q, *s, *sn, *m, *den, *num, *qsp, *ssp, *qsc, *ssc, *qn; int64_t* sy; int *nrun, *lab; unsigned long long* dbg;
    hipMalloc(&qsp, hq.size() * 4); hipMalloc(&ssp, hs.size() * 4); hipMalloc(&qsc, B * 4); hipMalloc(&ssc, N * 4); hipMalloc(&qn, B * 4);
    hipMalloc(&q, hq.size() * 4); hipMalloc(&s, hs.size() * 4); hipMalloc(&sn, N * 4); hipMalloc(&sy, N * 8);
    hipMalloc(&m, (size_t)n_stiles * B * 4); hipMalloc(&den, (size_t)n_stiles * B * 4); hipMalloc(&nrun, n_stiles * 4);
    hipMalloc(&lab, (size_t)n_stiles * BS * 4); hipMalloc(&num, (size_t)n_stiles * BS * B * 4); hipMalloc(&dbg, 1 << 22);
    hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice); hipMemcpy(s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(sy, hy.data(), N * 8, hipMemcpyHostToDevice); hipMemset(sn, 0, N * 4);
    launch_split_rows(q, qsp, qsc, qn, B, d, 0); launch_split_rows(s, ssp, ssc, sn, N, d, 0);
    // run tables of the persistent kernel, on the host
    std::vector<int> h_runid((size_t)n_stiles * BS + 64, 0), h_nrun(n_stiles, 0), h_lab((size_t)n_stiles * BS, -1), h_bnd((size_t)n_stiles * 2, BS);
    for (int stt = 0; stt < n_stiles; ++stt) {
        int id = -1; long long prev = -2;
        for (int t = 0; t < BS; ++t) {
            const int j = stt * BS + t;
            const long long y = j < N ? hy[j] : -1;
            if (t == 0 || y != prev) { ++id; h_lab[(size_t)stt * BS + id] = (int)y; if (id == 1 || id == 2) h_bnd[2 * stt + id - 1] = t; }
            prev = y;
            h_runid[(size_t)stt * BS + t] = id;
        }
        h_nrun[stt] = id + 1;
    }
    int* runid; hipMalloc(&runid, h_runid.size() * 4);
    int* bndp; hipMalloc(&bndp, h_bnd.size() * 4); hipMemcpy(bndp, h_bnd.data(), h_bnd.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(runid, h_runid.data(), h_runid.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(nrun, h_nrun.data(), h_nrun.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(lab, h_lab.data(), h_lab.size() * 4, hipMemcpyHostToDevice);
    FusedWs wsp; wsp.m = m; wsp.den = den; wsp.nrun = nrun; wsp.lab = lab; wsp.num = num; wsp.runid = runid; wsp.bnd = bndp;
    const int grid = padded_grid(n_stiles, n_qtiles);
    const size_t lds = FUSED_HDR + DmaCfg<RS>::STAGE_BYTES;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] {
        if (persistent)
            launch_f16p<RS, 0>(qsp, ssp, sy, sn, ssc, qn, qsc, nullptr, wsp, B, N, d, C, n_stiles, n_qtiles, 0);
        else if (qraw)
            hipLaunchKernelGGL((nw_fused_kernel<RS, 0, false, MODE_F16Q>), dim3(grid), dim3(TILE_THREADS), lds, 0, q, ssp, sy, sn, ssc,
                               (const float*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, (float*)dbg, m, den, nrun, lab, num, B, N, d, C, n_stiles, n_qtiles);
        else if (f16)
            hipLaunchKernelGGL((nw_fused_kernel<RS, 0, false, MODE_F16>), dim3(grid), dim3(TILE_THREADS), lds, 0, qsp, ssp, sy, sn, ssc, qn, qsc,
                               (const float*)nullptr, (float*)dbg, m, den, nrun, lab, num, B, N, d, C, n_stiles, n_qtiles);
        else
            hipLaunchKernelGGL((nw_fused_kernel<RS, 0, false, MODE_DMA_SN>), dim3(grid), dim3(TILE_THREADS), lds, 0, q, s, sy, sn, (const float*)nullptr,
                               (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)dbg, m, den, nrun, lab, num, B, N, d, C,
                               n_stiles, n_qtiles);
    };
    const int warm = argc > 6 ? atoi(argv[6]) : 5;   // e.g. 300: past the post-idle clock ramp (~40 ms)
    for (int i = 0; i < warm; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 100; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (persistent) {
        std::vector<unsigned long long> hp(8 * 1024);
        hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(nw_diag_p), hp.size() * 8);
        double ph[8] = {0}; int n = 0;
        const int nwg = persistent_variant() == 1 ? 512 : 256;
        for (int b = 0; b < nwg; ++b) { ++n; for (int k = 0; k < 8; ++k) ph[k] += (double)hp[8 * b + k]; }
        {   // the clock the kernel itself saw: shader ticks per 100 MHz real-time tick, per workgroup (LAST launch)
            std::vector<unsigned long long> rt(2 * 1024);
            hipMemcpyFromSymbol(rt.data(), HIP_SYMBOL(nw_diag_rt), rt.size() * 8);
            std::vector<double> ghz;
            for (int b = 0; b < nwg; ++b)
                if (rt[2 * b + 1] > rt[2 * b]) ghz.push_back((double)hp[8 * b + 7] / (double)(rt[2 * b + 1] - rt[2 * b]) * 0.1);
            std::sort(ghz.begin(), ghz.end());
            if (!ghz.empty())
                printf("in-kernel clock (ticks / s_memrealtime, per workgroup): median %.3f GHz, min %.3f, max %.3f; wave lifetime %.1f us\n",
                       ghz[ghz.size() / 2], ghz.front(), ghz.back(), ph[7] / n / (ghz[ghz.size() / 2] * 1e3));
        }
        printf("persistent kernel %.2f us | per WG (s_memtime ticks): main loops %.0f | epilogue: header reads %.0f, scores %.0f, mask+max+exp+den %.0f, run sums+stores %.0f, m/den stores %.0f, rest %.0f | total %.0f\n",
               ms * 10, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n, ph[6] / n, ph[7] / n);
        return 0;
    }
    std::vector<unsigned long long> h(8 * grid);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> hq2(4096);
    hipMemcpyFromSymbol(hq2.data(), HIP_SYMBOL(nw_diag_q), hq2.size() * 8);
    double ph[4] = {0}, ep[3] = {0}; int n = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < grid; ++b) {
        if (!h[8 * b + 6]) continue;
        ++n;
        ph[0] += (double)(h[8 * b + 1] - h[8 * b + 0]);
        ph[1] += (double)(h[8 * b + 2] - h[8 * b + 1]);
        ph[2] += (double)(h[8 * b + 6] - h[8 * b + 2]);
        if (qraw) ph[3] += (double)(hq2[b & 4095] - h[8 * b + 1]);
        ep[0] += (double)(h[8 * b + 3] - h[8 * b + 2]); ep[1] += (double)(h[8 * b + 4] - h[8 * b + 3]); ep[2] += (double)(h[8 * b + 6] - h[8 * b + 4]);
        tmin = std::min(tmin, h[8 * b + 0]); tmax = std::max(tmax, h[8 * b + 6]);
    }
    printf("kernel %.2f us | cycles per WG: scan+header %.0f, main loop (incl. query prologue %.0f) %.0f, epilogue %.0f | first start -> last end %.0f ticks (n=%d)\n",
           ms * 10, ph[0] / n, ph[3] / n, ph[1] / n, ph[2] / n, (double)(tmax - tmin), n);
    printf("   epilogue: scores+max %.0f, exp2 + run sums + num stores %.0f, m/den stores + run table + end %.0f\n", ep[0] / n, ep[1] / n, ep[2] / n);
    return 0;
}
