// fused.hip -- the fused NW-head forward for a shared (N,d) support (gfx950 / MI355X only).
//
// Replaces, in ONE pass over the support rows and without materialising the (B,N) score matrix,
// the reference sequence  one_hot -> kernel scores -> softmax -> bmm -> log  (nwhead/nw.py:276-289,
// scores from nwhead/kernel.py:13-44):
//
//   nw_fused_kernel   per workgroup tile (64 queries x 16*RS supports): fp32-MFMA dot products
//                     (tile_core.h), score epilogue, tile-local softmax statistics
//                     m = max_j s, den = sum_j e^(s-m), and the e^(s-m) sums of every RUN of equal
//                     consecutive labels inside the tile (a class-sorted bank has 1-2 runs per tile;
//                     an unsorted one degrades gracefully to one run per support).
//   nw_merge_runs_kernel  per query: rescale every tile to the global max, add run sums into the
//                     per-class accumulator, write log(num/den + 1e-12) (or the (m, den, num)
//                     partials of a shard).
#include <mutex>
#include <utility>
#include <vector>
#include "fused_impl.h"

namespace nw {

// instantiated in fused_k0.hip .. fused_k4.hip
#define NW_EXTERN_FUSED_KIND(K)                                                                          \
    extern template int launch_fused_kind<K>(const float*, const float*, const int64_t*, const float*,   \
                                             const float*, const float*, const float*,                   \
                                             const float*, float*, float*, float*, float*, float*,       \
                                             float*, void*, size_t, int, int, int, int, hipStream_t);
NW_EXTERN_FUSED_KIND(NW_SCORE_EUCLIDEAN)
NW_EXTERN_FUSED_KIND(NW_SCORE_HYPERSPHERE)
NW_EXTERN_FUSED_KIND(NW_SCORE_COSINE)
NW_EXTERN_FUSED_KIND(NW_SCORE_DOT)
NW_EXTERN_FUSED_KIND(NW_SCORE_CLIP)
#undef NW_EXTERN_FUSED_KIND

// ---- diagnostics: device time of the tile kernel alone (nw_debug_tile_timing*, include/nwhead_hip.h)
namespace {
struct TileTimer {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;  // pairs recorded since the last read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    std::mutex mu;
} g_tt;
}  // namespace

int tile_timer_start(hipStream_t st) {
    if (!g_tt.on) return -1;
    std::lock_guard<std::mutex> lk(g_tt.mu);
    std::pair<hipEvent_t, hipEvent_t> p;
    if (!g_tt.pool.empty()) {
        p = g_tt.pool.back();
        g_tt.pool.pop_back();
    } else if (hipEventCreate(&p.first) != hipSuccess || hipEventCreate(&p.second) != hipSuccess) {
        return -1;
    }
    (void)hipEventRecord(p.first, st);
    g_tt.ev.push_back(p);
    return (int)g_tt.ev.size() - 1;
}
void tile_timer_stop(int slot, hipStream_t st) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_tt.mu);
    if (slot < (int)g_tt.ev.size()) (void)hipEventRecord(g_tt.ev[slot].second, st);
}
int tile_timer_enable(bool on) {
    std::lock_guard<std::mutex> lk(g_tt.mu);
    g_tt.on = on;
    return NW_OK;
}
int tile_timer_read(double* total_us, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_tt.mu);
    double tot = 0;
    for (auto& p : g_tt.ev) {
        float ms = 0.f;
        if (hipEventSynchronize(p.second) != hipSuccess || hipEventElapsedTime(&ms, p.first, p.second) != hipSuccess)
            return NW_ERR_LAUNCH;
        tot += (double)ms * 1e3;
        g_tt.pool.push_back(p);
    }
    if (total_us) *total_us = tot;
    if (launches) *launches = (int64_t)g_tt.ev.size();
    g_tt.ev.clear();
    return NW_OK;
}

int persistent_variant() {
    static int v = [] {
        const char* e = getenv("NW_PVAR");
        const int x = e ? atoi(e) : -1;
        return (x >= 0 && x <= 2) ? x : -1;  // -1: chosen per launch
    }();
    return v;
}

int persistent_qgroup() {
    static int v = [] {
        const char* e = getenv("NW_QG");
        const int x = e ? atoi(e) : 0;
        return (x >= 1 && x <= 64) ? x : 8;
    }();
    return v;
}

int device_cu_count() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}
bool env_flag(const char* name) {
    const char* e = getenv(name);
    return e && e[0] == '1';
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t fused_layout(int64_t B, int64_t n_stiles, int BS, char* base, FusedWs* ws) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += al256(bytes);
        return p;
    };
    FusedWs w;
    w.m = reinterpret_cast<float*>(take((size_t)n_stiles * B * 4));
    w.den = reinterpret_cast<float*>(take((size_t)n_stiles * B * 4));
    w.nrun = reinterpret_cast<int*>(take((size_t)n_stiles * 4));
    w.lab = reinterpret_cast<int*>(take((size_t)n_stiles * BS * 4));
    w.num = reinterpret_cast<float*>(take((size_t)n_stiles * BS * B * 4));
    w.runid = reinterpret_cast<int*>(take(((size_t)n_stiles * BS + 64) * 4));
    w.bnd = reinterpret_cast<int*>(take((size_t)n_stiles * 2 * 4));
    if (ws) *ws = w;
    return off;
}

namespace {

// One workgroup per query.  Tiles are rescaled to the global max M; run sums go to their class.
// ws_m is in base-2 units (u = score * log2 e, see nw_fused_kernel).
template <bool PARTIAL>
__global__ __launch_bounds__(256) void nw_merge_runs_kernel(
    const float* __restrict__ ws_m, const float* __restrict__ ws_den, const int* __restrict__ ws_nrun,
    const int* __restrict__ ws_lab, const float* __restrict__ ws_num, float* __restrict__ out,
    float* __restrict__ lse, float* __restrict__ m_out, float* __restrict__ den_out,
    float* __restrict__ num_out, int B, int C, int n_stiles, int BS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    float* num = red + 8;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) num[c] = 0.f;

    float M = -INFINITY;
    for (int t = tid; t < n_stiles; t += 256) M = fmaxf(M, ws_m[(size_t)t * B + b]);
    M = block_max(M, red);

    float den = 0.f;
    for (int t = tid; t < n_stiles; t += 256) {
        const float f = __builtin_amdgcn_exp2f(ws_m[(size_t)t * B + b] - M);
        den += ws_den[(size_t)t * B + b] * f;
        const int nr = ws_nrun[t];
        for (int r = 0; r < nr; ++r) {
            const int y = ws_lab[(size_t)t * BS + r];
            if (y >= 0) atomicAdd(&num[y], ws_num[((size_t)t * BS + r) * B + b] * f);
        }
    }
    den = block_sum(den, red);  // its barriers also order the LDS adds before the reads below
    __syncthreads();
    if (PARTIAL) {
        if (tid == 0) {
            m_out[b] = M * 0.693147180559945309417f;  // back to natural units
            den_out[b] = den;
        }
        for (int c = tid; c < C; c += 256) num_out[(size_t)b * C + c] = num[c];
    } else {
        const float inv = 1.f / den;
        if (tid == 0 && lse) lse[b] = M * 0.693147180559945309417f + logf(den);
        for (int c = tid; c < C; c += 256) out[(size_t)b * C + c] = logf(num[c] * inv + NW_LOG_EPS);
    }
}

// (Merging inside the tile kernel -- the last workgroup of a query tile does it -- was tried and dropped:
//  one CU's ~450 dependent-latency loads take 25 us where this kernel's 256 workgroups take 6, and
//  agent-scope release/acquire fences cost ~40 us per launch on the 8-XCD part.)
// The same merge for large query batches: one workgroup per MQ = 16 consecutive queries, its 256
// threads = 16 queries x 16 tile lanes, so every workspace read is a 64-byte run along the query
// axis (the one-workgroup-per-query kernel above reads 4 bytes per 16 KB stride: 53 us at B = 4096,
// n_stiles = 391).  Class sums live in LDS as num[class][query].
// MQ x ML = 16 x 32 when there are many tiles per query (K3: 391); 32 x 16 when there are few (a shard of
// the bank at 8 ranks: 49 tiles -- measured there 14.9 / 11.1 / 11.5 us for MQ = 16 / 32 / 64).
// (A query-blocked workspace with packed run rows -- one contiguous 40 KB piece per 128 queries instead
//  of 256-byte runs B*4 bytes apart -- was measured too: no change; the walk is bound by its dependent
//  load latencies, not by locality.)
constexpr int MTHREADS = 512, MU = 4;
template <bool PARTIAL, int MQ>
__global__ __launch_bounds__(MTHREADS) void nw_merge_runs_blk_kernel(
    const float* __restrict__ ws_m, const float* __restrict__ ws_den, const int* __restrict__ ws_nrun,
    const int* __restrict__ ws_lab, const float* __restrict__ ws_num, float* __restrict__ out,
    float* __restrict__ lse, float* __restrict__ m_out, float* __restrict__ den_out,
    float* __restrict__ num_out, int B, int C, int n_stiles, int BS) {
    constexpr int ML = MTHREADS / MQ;
    constexpr int MNS = MQ + 1;  // row stride of num[class][query] in LDS: odd, the output phase reads columns
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [ML][MQ]
    float* inv_s = red + ML * MQ;                 // [MQ]
    float* num = inv_s + MQ;                      // [C][MQ]
    const int tid = threadIdx.x, bq = tid & (MQ - 1), sl = tid / MQ;
    const int b0 = blockIdx.x * MQ;
    const int b = min(b0 + bq, B - 1);  // rows past the batch repeat the last query and are never written
    for (int x = tid; x < C * MNS; x += MTHREADS) num[x] = 0.f;

    float M = -INFINITY;
    for (int t = sl; t < n_stiles; t += ML) M = fmaxf(M, ws_m[(size_t)t * B + b]);
    red[sl * MQ + bq] = M;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ML; ++k) M = fmaxf(M, red[k * MQ + bq]);
    __syncthreads();

    // The walk is latency-bound (a tile's run count gates its label and sum reads), so MU tiles are in
    // flight per thread and the first two runs of each are read before the count is known: rows past
    // nrun are allocated but never written, so what comes back is selected away, never used in arithmetic.
    float den = 0.f;
    for (int t0 = sl; t0 < n_stiles; t0 += ML * MU) {
        float mm[MU], dd[MU], n0[MU], n1[MU];
        int nr[MU], y0[MU], y1[MU], tt[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int t = t0 + u * ML;
            const bool ok = t < n_stiles;
            const int tc = ok ? t : t0;
            tt[u] = tc;
            mm[u] = ws_m[(size_t)tc * B + b];
            dd[u] = ws_den[(size_t)tc * B + b];
            nr[u] = ok ? ws_nrun[tc] : -1;
            y0[u] = ws_lab[(size_t)tc * BS];
            y1[u] = ws_lab[(size_t)tc * BS + 1];
            n0[u] = ws_num[((size_t)tc * BS) * B + b];
            n1[u] = ws_num[((size_t)tc * BS + 1) * B + b];
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            if (nr[u] < 0) continue;
            const float f = __builtin_amdgcn_exp2f(mm[u] - M);
            den += dd[u] * f;
            if (nr[u] > 0 && y0[u] >= 0) atomicAdd(&num[y0[u] * MNS + bq], n0[u] * f);
            if (nr[u] > 1 && y1[u] >= 0) atomicAdd(&num[y1[u] * MNS + bq], n1[u] * f);
            for (int r = 2; r < nr[u]; ++r) {
                const int y = ws_lab[(size_t)tt[u] * BS + r];
                if (y >= 0) atomicAdd(&num[y * MNS + bq], ws_num[((size_t)tt[u] * BS + r) * B + b] * f);
            }
        }
    }
    red[sl * MQ + bq] = den;
    __syncthreads();  // also orders the LDS adds before the reads below
    den = 0.f;
#pragma unroll
    for (int k = 0; k < ML; ++k) den += red[k * MQ + bq];
    const int nq = min(MQ, B - b0);
    if (PARTIAL) {
        if (sl == 0 && bq < nq) {
            m_out[b] = M * 0.693147180559945309417f;  // back to natural units
            den_out[b] = den;
        }
        for (int x = tid; x < nq * C; x += MTHREADS) {
            const int qq = x / C, c = x - qq * C;
            num_out[(size_t)(b0 + qq) * C + c] = num[c * MNS + qq];
        }
    } else {
        if (sl == 0) {
            inv_s[bq] = 1.f / den;
            if (lse && bq < nq) lse[b] = M * 0.693147180559945309417f + logf(den);
        }
        __syncthreads();
        for (int x = tid; x < nq * C; x += MTHREADS) {
            const int qq = x / C, c = x - qq * C;
            out[(size_t)(b0 + qq) * C + c] = logf(num[c * MNS + qq] * inv_s[qq] + NW_LOG_EPS);
        }
    }
}

// Runs of equal consecutive labels inside every support tile of BS rows, one wave per tile:
// runid[st*BS + t] = run of tile row t, lab[st*BS + run] = its class (-1: padding rows / labels outside
// [0, C)), nrun[st].  The persistent kernel reads these instead of scanning the labels once per
// (query tile, support tile) pair.
__global__ __launch_bounds__(64) void nw_run_tables_kernel(const int64_t* __restrict__ sy, int N, int C, int BS,
                                                           int* __restrict__ runid, int* __restrict__ nrun,
                                                           int* __restrict__ lab_out, int* __restrict__ bnd) {
    const int st = blockIdx.x, lane = threadIdx.x, s0 = st * BS;
    if (lane < 2) bnd[2 * st + lane] = BS;  // overwritten below when runs 1 / 2 exist (same lane order: see the barrier)
    __syncthreads();
    int lab[3], flag[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = 3 * lane + u, j = s0 + t;
        int64_t y = -1;
        if (t < BS && j < N) y = sy[j];
        lab[u] = ((uint64_t)y < (uint64_t)C) ? (int)y : -1;
    }
    const int prev_last = __shfl_up(lab[2], 1);
    flag[0] = (lane == 0) || (lab[0] != prev_last);
    flag[1] = lab[1] != lab[0];
    flag[2] = lab[2] != lab[1];
    int incl = flag[0] + flag[1] + flag[2];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    int id = incl - (flag[0] + flag[1] + flag[2]) - 1;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = 3 * lane + u;
        id += flag[u];
        if (t < BS) {
            runid[(size_t)s0 + t] = id;
            if (flag[u]) lab_out[(size_t)s0 + id] = lab[u];
            if (flag[u] && (id == 1 || id == 2)) bnd[2 * st + id - 1] = t;
            if (t == BS - 1) nrun[st] = id + 1;
        }
    }
}

int env_rs() {
    static int v = [] {
        const char* e = getenv("NW_TILE_RS");
        return e ? atoi(e) : 0;
    }();
    return v;
}

}  // namespace

int launch_run_tables(const FusedWs& ws, const int64_t* sy, int N, int C, int n_stiles, int BS, hipStream_t st) {
    if (BS > 192) return NW_ERR_UNSUPPORTED;  // three rows per lane
    hipLaunchKernelGGL(nw_run_tables_kernel, dim3(n_stiles), dim3(64), 0, st, sy, N, C, BS, ws.runid, ws.nrun, ws.lab, ws.bnd);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

int launch_merge_runs(const FusedWs& ws, float* out, float* lse, float* m, float* den, float* num,
                      int B, int C, int n_stiles, int BS, hipStream_t st) {
    // 32 queries x 16 tile lanes per workgroup when a query has few tiles (a shard of the bank), else 16 x 32
    const int mq = (n_stiles >= 128 || ((size_t)MTHREADS + 32 + (size_t)C * 33) * sizeof(float) > 64 * 1024) ? 16 : 32;
    const size_t blds = ((size_t)MTHREADS + mq + (size_t)C * (mq + 1)) * sizeof(float);
    if (B >= 512 && blds <= 64 * 1024 && !env_flag("NW_MERGE_PER_QUERY")) {
        const int grid = (B + mq - 1) / mq;
#define NW_MERGE_BLK(P_, Q_)                                                                                          \
    hipLaunchKernelGGL((nw_merge_runs_blk_kernel<P_, Q_>), dim3(grid), dim3(MTHREADS), blds, st, ws.m, ws.den, ws.nrun, \
                       ws.lab, ws.num, out, lse, m, den, num, B, C, n_stiles, BS)
        if (out) { if (mq == 16) NW_MERGE_BLK(false, 16); else NW_MERGE_BLK(false, 32); }
        else { if (mq == 16) NW_MERGE_BLK(true, 16); else NW_MERGE_BLK(true, 32); }
#undef NW_MERGE_BLK
        NW_CHECK_LAUNCH();
        return NW_OK;
    }
    const size_t mlds = (8 + (size_t)C) * sizeof(float);
    if (out)
        hipLaunchKernelGGL(nw_merge_runs_kernel<false>, dim3(B), dim3(256), mlds, st, ws.m, ws.den, ws.nrun,
                           ws.lab, ws.num, out, lse, m, den, num, B, C, n_stiles, BS);
    else
        hipLaunchKernelGGL(nw_merge_runs_kernel<true>, dim3(B), dim3(256), mlds, st, ws.m, ws.den, ws.nrun,
                           ws.lab, ws.num, out, lse, m, den, num, B, C, n_stiles, BS);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

// Support-tile height (in 16-row blocks).  Model: workgroups run in rounds of 256 (one per CU at
// full MFMA rate; two co-resident ones share the pipe but fill each other's barrier bubbles, worth
// ~15 %), each costs RS blocks of MFMA work plus a fixed prologue/epilogue.
int pick_rs(int64_t B, int64_t N, int64_t d, bool f16) {
    const int forced = env_rs();
    if (forced == 2 || forced == 4 || forced == 5 || forced == 6 || forced == 8 || forced == 10 || forced == 12)
        return forced;
    const int64_t nq = (B + BQ - 1) / BQ;
    // LDS-DMA path (d % 32 == 0): 80-row tiles keep a workgroup under 80 KB of LDS and 128 VGPRs, so
    // two share a CU and hide each other's prologue, barriers and epilogue (measured 111.6 vs 105.2
    // TFLOP/s at B=2048 N=50000, 36.2 vs 37.0 us at T) -- worth it once they fill the 512 slots.
    // (the split-fp16 path is bound by the L2->LDS stream, not by the matrix pipe: the larger tile,
    //  which moves fewer bytes per flop, wins there: 470 vs 509 us at B=2048 N=50000)
    if (forced == 0 && !f16 && d % BK == 0 && nq * ((N + 79) / 80) >= 480) return 5;
    // split-fp16 with >= 4 tiles per CU: the persistent kernel (fused_f16p.h) at its no-spill height
    if (forced == 0 && f16 && d % BK == 0 && nq * ((N + 127) / 128) >= 4 * 256) return 8;
    const int cand[] = {2, 4, 6, 8, 10, 12};  // even: the four loader waves split a tile evenly
    const int ncand = f16 ? 5 : 6;           // split-fp16: 12 blocks of fragments do not fit 256 VGPRs
    double best = 1e30;
    int best_rs = 8;
    for (int ci = 0; ci < ncand; ++ci) {
        const int rs = cand[ci];
        const int64_t ns = (N + 16 * rs - 1) / (16 * rs);
        const int64_t nwg = nq * ns;
        const int64_t rounds = (nwg + 255) / 256;
        double cost = (double)rounds * (rs + 1.5);
        if (nwg >= 384) cost *= 0.85;
        if (cost < best - 1e-9) {
            best = cost;
            best_rs = rs;
        }
    }
    return best_rs;
}

size_t fused_workspace_bytes(int64_t B, int64_t N, int64_t d) {
    size_t need = 0;
    for (int f16 = 0; f16 < 2; ++f16) {  // either operand form may be chosen at launch
        const int rs = pick_rs(B, N, d, f16 != 0);
        const int64_t n_stiles = (N + 16 * rs - 1) / (16 * rs);
        const size_t n = fused_layout(B, n_stiles, 16 * rs, nullptr, nullptr);
        need = n > need ? n : need;
    }
    return need;
}

bool fused_eligible(const float* q, const float* s, int64_t B, int64_t N, int64_t d, int64_t C) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(s)) & 15) == 0;
    return N > 25 && d >= 4 && (d % 4) == 0 && aligned && B < (1 << 30) && N < (1 << 30) &&
           d < (1 << 30) && C < (1 << 30) && (8 + C) * 4 <= 160 * 1024;
}

// out != nullptr: final log-probabilities (+ optional scores / lse); out == nullptr: (m, den, num).
int launch_fused(const float* q, const float* s, const int64_t* sy, const float* s_norm2,
                 const float* s_scale, const float* q_norm2, const float* q_scale, const float* ls, float* out,
                 float* scores, float* lse, float* m, float* den, float* num, void* workspace,
                 size_t workspace_bytes, int64_t B, int64_t N, int64_t d, int64_t C, int kind,
                 hipStream_t st) {
#define NW_KIND_CASE(K) \
    case K: return launch_fused_kind<K>(q, s, sy, s_norm2, s_scale, q_norm2, q_scale, ls, out, scores, lse, m, den, num, workspace, workspace_bytes, (int)B, (int)N, (int)d, (int)C, st)
    switch (kind) {
        NW_KIND_CASE(NW_SCORE_EUCLIDEAN);
        NW_KIND_CASE(NW_SCORE_HYPERSPHERE);
        NW_KIND_CASE(NW_SCORE_COSINE);
        NW_KIND_CASE(NW_SCORE_DOT);
        NW_KIND_CASE(NW_SCORE_CLIP);
        default: return NW_ERR_UNSUPPORTED;
    }
#undef NW_KIND_CASE
}

}  // namespace nw
