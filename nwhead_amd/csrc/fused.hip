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
#include <cstring>

namespace nw {

// instantiated in fused_k0.hip .. fused_k4.hip
#define NW_EXTERN_FUSED_KIND(K)                                                                          \
    extern template int launch_fused_kind<K>(const float*, const float*, const int64_t*, const float*,   \
                                             const float*,                                               \
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
    const int x = knob(KNOB_PVAR);
    return (x >= 0 && x <= 2) ? x : -1;  // -1: chosen per launch
}

int persistent_qgroup() {
    const int x = knob(KNOB_QG);
    return (x >= 1 && x <= 64) ? x : 8;
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
bool env_flag(const char* name) {   // (named after the variables the Python layer forwards: diagnostic knobs)
    static const struct { const char* n; int k; } map[] = {
        {"NW_MERGE_PER_QUERY", KNOB_MERGE_PER_QUERY}, {"NW_MERGE_NO_GLOBAL_TABLES", KNOB_MERGE_NO_GLOBAL_TABLES},
        {"NW_PERSISTENT_ANY_RS", KNOB_PERSISTENT_ANY_RS}, {"NW_NO_PERSISTENT", KNOB_NO_PERSISTENT}};
    for (const auto& m : map)
        if (!strcmp(m.n, name)) return knob(m.k) == 1;
    return false;
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t fused_layout(int64_t B, int64_t n_stiles, int BS, char* base, FusedWs* ws, int64_t C) {
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
    w.ctab = reinterpret_cast<int*>(take((size_t)C * (3 + 4) * 4));   // class tables of the run merge when they do not fit in LDS
    if (ws) *ws = w;
    return off;
}

namespace {

// The run merge.  One workgroup per MQ consecutive queries; ws_m is in base-2 units (u = score * log2 e,
// see nw_fused_kernel).  Three phases, two global round trips, NO float atomics -- every sum has a fixed
// order, so the output is bit-reproducible from launch to launch:
//   0  (tables, shared by the workgroup's queries) per class c: the (tile, run) entries that carry it, in
//      bank order.  A class-sorted bank gives a class 1-3 entries; up to MENT are kept in LDS (integer LDS
//      atomics hand out the slots, the slots are then sorted: deterministic), a class with more entries is
//      found again by scanning the run labels of tiles [tlo[c], thi[c]].
//   1  M = max over tiles, den = sum_t den_t 2^(m_t - M): threads = MQ queries x ML tile lanes, each lane
//      sums its tiles in order, the lanes are added in order.
//   2  thread per (class, query): its entries' run sums, rescaled, added in bank order; every address is
//      known from the LDS tables, so the loads of one thread fly together.
// The reads of phases 0 and 1 do not depend on each other: one round trip; phase 2 is the second.
// (History: one LDS float atomicAdd per (tile, run) was 6.2 us at T and not reproducible; merging inside
//  the tile kernel -- the last workgroup of a query tile does it -- was tried and dropped: one CU's ~450
//  dependent-latency loads take 25 us, and agent-scope release/acquire fences cost ~40 us per launch on
//  the 8-XCD part.  A query-blocked workspace with packed run rows was measured too: no change.)
constexpr int MTHREADS = 512, MENT = 4, MTPT = 16;
// TMODE: where the per-class tables (count, first / last tile, up to MENT entries) live: 1 = LDS, built by every
// workgroup (they fit up to a few thousand classes); 2 = global memory, built once per launch by
// nw_class_tables_kernel (more classes than LDS holds: MQ = 1, results written directly); 0 = none (every class scans
// every tile: only when there is no workspace for mode 2).
template <bool PARTIAL, int MQ, int TMODE>
__global__ __launch_bounds__(MTHREADS) void nw_merge_runs_kernel(
    const float* __restrict__ ws_m, const float* __restrict__ ws_den, const int* __restrict__ ws_nrun,
    const int* __restrict__ ws_lab, const float* __restrict__ ws_num, float* __restrict__ out,
    float* __restrict__ lse, float* __restrict__ m_out, float* __restrict__ den_out,
    float* __restrict__ num_out, int B, int C, int n_stiles, int BS, const int* __restrict__ ctab) {
    constexpr bool TABLES = TMODE != 0;
    constexpr bool LTAB = TMODE == 1;   // tables built here, in LDS
    constexpr int ML = MTHREADS / MQ;  // tile lanes per query
    constexpr int MNS = MQ + 1;        // row stride of res[class][query] in LDS: odd, the output phase reads columns
    constexpr int NW = MTHREADS / 64;  // waves
    constexpr float LN2 = 0.693147180559945309417f;
    static_assert(MQ <= 32 && (MQ & (MQ - 1)) == 0, "a wave holds whole groups of MQ queries");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [NW][MQ] x 2
    float* Ms = red + 2 * NW * MQ;                // [MQ]
    float* inv_s = Ms + MQ;                       // [MQ]
    int* lds_tab = reinterpret_cast<int*>(inv_s + MQ);
    int* cnt = TMODE == 2 ? const_cast<int*>(ctab) : lds_tab;  // [C]   entries of class c   (TABLES only, like the next four)
    int* tlo = cnt + C;                             // [C]   first tile carrying c
    int* thi = tlo + C;                             // [C]   last tile carrying c
    int* ent = thi + C;                             // [C][MENT]  tile * BS + run, ascending
    float* res = reinterpret_cast<float*>(lds_tab + (size_t)C * (3 + MENT));  // [C][MNS]   (LDS tables, MQ > 1)
    const int tid = threadIdx.x, bq = tid & (MQ - 1), sl = tid / MQ, wave = tid >> 6;
    const int b0 = blockIdx.x * MQ;
    const int b = min(b0 + bq, B - 1);  // rows past the batch repeat the last query and are never written

    // ---- round trip 1: this thread's tiles (kept in registers when there are at most MTPT of them) and the
    // run tables.  The first three run labels of a tile are read before its run count is known (rows past nrun
    // are allocated, never written: what comes back is selected away).
    const bool in_regs = n_stiles <= ML * MTPT;
    float mr[MTPT], dr[MTPT];
    float M = -INFINITY;
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < MTPT; ++u) {
            const int t = sl + u * ML;
            const bool ok = t < n_stiles;
            mr[u] = ok ? ws_m[(size_t)t * B + b] : -INFINITY;
            dr[u] = ok ? ws_den[(size_t)t * B + b] : 0.f;
        }
    }
    int tnr = 0, tl3[3] = {-1, -1, -1};
    const bool one_pass_tables = LTAB && n_stiles <= MTHREADS;
    if (one_pass_tables && tid < n_stiles) {
        tnr = ws_nrun[tid];
        const int* lt = ws_lab + (size_t)tid * BS;
        tl3[0] = lt[0];
        tl3[1] = lt[1];
        tl3[2] = lt[2];
    }
    if (LTAB) {
        for (int c = tid; c < C; c += MTHREADS) {
            cnt[c] = 0;
            tlo[c] = 0x7fffffff;
            thi[c] = -1;
        }
    }
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < MTPT; ++u) M = fmaxf(M, mr[u]);
    } else {
        for (int t = sl; t < n_stiles; t += ML) M = fmaxf(M, ws_m[(size_t)t * B + b]);
    }
#pragma unroll
    for (int o = 32; o >= MQ; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));  // lanes l, l ^ MQ, ... hold the same query
    if ((tid & 63) < MQ) red[wave * MQ + bq] = M;
    __syncthreads();  // also publishes the reset tables
    // ---- phase 0: class tables
    if (LTAB) {
        auto add_tile = [&](int t, int nr, const int (&l3)[3]) {
            const int* lt = ws_lab + (size_t)t * BS;
            for (int r = 0; r < nr; ++r) {
                const int y = r < 3 ? l3[r] : lt[r];
                if (y < 0) continue;  // padding rows / labels outside [0, C)
                atomicMin(&tlo[y], t);
                atomicMax(&thi[y], t);
                const int slot = atomicAdd(&cnt[y], 1);
                if (slot < MENT) ent[y * MENT + slot] = t * BS + r;
            }
        };
        if (one_pass_tables) {
            if (tid < n_stiles) add_tile(tid, tnr, tl3);
        } else {
            for (int t = tid; t < n_stiles; t += MTHREADS) {
                const int* lt = ws_lab + (size_t)t * BS;
                const int l3[3] = {lt[0], lt[1], lt[2]};
                add_tile(t, ws_nrun[t], l3);
            }
        }
    }
    M = red[bq];
#pragma unroll
    for (int k = 1; k < NW; ++k) M = fmaxf(M, red[k * MQ + bq]);
    // ---- den: every thread adds its tiles in order, the tile lanes of a wave are added in a fixed tree, the
    // waves in order
    float den = 0.f;
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < MTPT; ++u) den += dr[u] * __builtin_amdgcn_exp2f(mr[u] - M);  // absent tiles: 0 * 2^-inf = 0
    } else {
        for (int t = sl; t < n_stiles; t += ML)
            den += ws_den[(size_t)t * B + b] * __builtin_amdgcn_exp2f(ws_m[(size_t)t * B + b] - M);
    }
#pragma unroll
    for (int o = 32; o >= MQ; o >>= 1) den += __shfl_xor(den, o);
    if ((tid & 63) < MQ) red[NW * MQ + wave * MQ + bq] = den;
    if (sl == 0) Ms[bq] = M;
    __syncthreads();  // partial dens and the class tables are complete
    if (LTAB) {  // slots were handed out in arrival order: put each class's entries in bank order
        for (int c = tid; c < C; c += MTHREADS) {
            const int n = cnt[c];
            if (n > 1 && n <= MENT) {
                int e[MENT];
#pragma unroll
                for (int k = 0; k < MENT; ++k) e[k] = k < n ? ent[c * MENT + k] : 0x7fffffff;
#pragma unroll
                for (int i = 1; i < MENT; ++i)
#pragma unroll
                    for (int j = MENT - 1; j >= i; --j)
                        if (e[j] < e[j - 1]) { const int x = e[j]; e[j] = e[j - 1]; e[j - 1] = x; }
#pragma unroll
                for (int k = 0; k < MENT; ++k)
                    if (k < n) ent[c * MENT + k] = e[k];
            }
        }
    }
    den = red[NW * MQ + bq];
#pragma unroll
    for (int k = 1; k < NW; ++k) den += red[NW * MQ + k * MQ + bq];
    if (sl == 0) inv_s[bq] = 1.f / den;
    const int nq = min(MQ, B - b0);
    if (sl == 0 && bq < nq) {
        if (PARTIAL) {
            m_out[b] = M * LN2;  // back to natural units
            den_out[b] = den;
        } else if (lse) {
            lse[b] = M * LN2 + logf(den);
        }
    }
    __syncthreads();  // sorted entries, inv_s
    // ---- round trip 2: class sums.  Few classes (C * MQ well under the workgroup's 512 threads) mean many tiles per
    // class: P lanes then share one (class, query) item -- lane `sub` takes the tiles t0 + sub, t0 + sub + P, ... and the
    // P partial sums are added in a fixed shuffle tree (a single class over 63 tiles: 36 -> 9 us of merge at T).
    int P = 1;
    while (P < 64 && 2 * P * C * MQ <= MTHREADS) P <<= 1;
    for (int xx = tid; xx < C * MQ * P; xx += MTHREADS) {
        const int x = xx / P, sub = xx - x * P;
        const int c = x / MQ, q = x - c * MQ;
        const int bb = min(b0 + q, B - 1);
        const float Mq = Ms[q];
        float acc = 0.f;
        const int n = TABLES ? cnt[c] : MENT + 1;
        if (n <= MENT) {
            int e[MENT];
            float v[MENT], mt[MENT];
#pragma unroll
            for (int k = 0; k < MENT; ++k) e[k] = ent[c * MENT + (k < n ? k : 0)];
#pragma unroll
            for (int k = 0; k < MENT; ++k) {
                if (k < n) {
                    v[k] = ws_num[(size_t)e[k] * B + bb];
                    mt[k] = ws_m[(size_t)(e[k] / BS) * B + bb];
                }
            }
#pragma unroll
            for (int k = 0; k < MENT; ++k)
                if (k < n) acc += v[k] * __builtin_amdgcn_exp2f(mt[k] - Mq);
        } else {
            const int t0 = TABLES ? tlo[c] : 0, t1 = TABLES ? thi[c] : n_stiles - 1;
            for (int t = t0 + sub; t <= t1; t += P) {
                const int nr = ws_nrun[t];
                const float f = __builtin_amdgcn_exp2f(ws_m[(size_t)t * B + bb] - Mq);
                const int* lt = ws_lab + (size_t)t * BS;
                for (int r = 0; r < nr; ++r)
                    if (lt[r] == c) acc += ws_num[((size_t)t * BS + r) * B + bb] * f;
            }
        }
        if (P > 1) {   // (every lane of the item's group is here: groups are whole, the loop bounds are theirs alike)
            if (n <= MENT && sub != 0) acc = 0.f;
            for (int o = P >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (sub != 0) continue;
        }
        if (LTAB && MQ > 1) {
            res[c * MNS + q] = acc;
        } else if (q < nq) {  // one query per workgroup (consecutive classes: coalesced), or no room for the staging table
            if (PARTIAL) num_out[(size_t)bb * C + c] = acc;
            else out[(size_t)bb * C + c] = logf(acc * inv_s[q] + NW_LOG_EPS);
        }
    }
    if (!LTAB || MQ == 1) return;
    __syncthreads();
    for (int x = tid; x < nq * C; x += MTHREADS) {
        const int qq = x / C, c = x - qq * C;
        if (PARTIAL) num_out[(size_t)(b0 + qq) * C + c] = res[c * MNS + qq];
        else out[(size_t)(b0 + qq) * C + c] = logf(res[c * MNS + qq] * inv_s[qq] + NW_LOG_EPS);
    }
}

// The run merge's per-class tables in GLOBAL memory (ctab = cnt[C] | tlo[C] | thi[C] | ent[C][MENT]), for class counts
// whose tables do not fit in LDS: built once per launch by ONE workgroup (the tables depend on the labels and the
// tiling only, not on the queries), integer atomics hand out the slots, then every class's entries are put in bank
// order -- the same construction nw_merge_runs_kernel does per workgroup in LDS.
__global__ __launch_bounds__(1024) void nw_class_tables_kernel(const int* __restrict__ ws_nrun, const int* __restrict__ ws_lab,
                                                                int n_stiles, int BS, int C, int* __restrict__ ctab) {
    int* cnt = ctab;
    int* tlo = cnt + C;
    int* thi = tlo + C;
    int* ent = thi + C;
    const int tid = threadIdx.x;
    for (int c = tid; c < C; c += 1024) {
        cnt[c] = 0;
        tlo[c] = 0x7fffffff;
        thi[c] = -1;
    }
    __threadfence();
    __syncthreads();
    for (int t = tid; t < n_stiles; t += 1024) {
        const int nr = ws_nrun[t];
        const int* lt = ws_lab + (size_t)t * BS;
        for (int r = 0; r < nr; ++r) {
            const int y = lt[r];
            if (y < 0) continue;
            atomicMin(&tlo[y], t);
            atomicMax(&thi[y], t);
            const int slot = atomicAdd(&cnt[y], 1);
            if (slot < MENT) ent[y * MENT + slot] = t * BS + r;
        }
    }
    __threadfence();
    __syncthreads();
    for (int c = tid; c < C; c += 1024) {
        const int n = cnt[c];
        if (n > 1 && n <= MENT) {
            int e[MENT];
#pragma unroll
            for (int k = 0; k < MENT; ++k) e[k] = k < n ? ent[c * MENT + k] : 0x7fffffff;
#pragma unroll
            for (int i = 1; i < MENT; ++i)
#pragma unroll
                for (int j = MENT - 1; j >= i; --j)
                    if (e[j] < e[j - 1]) { const int x = e[j]; e[j] = e[j - 1]; e[j - 1] = x; }
#pragma unroll
            for (int k = 0; k < MENT; ++k)
                if (k < n) ent[c * MENT + k] = e[k];
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
    const int v = knob(KNOB_TILE_RS);
    return v == KNOB_UNSET ? 0 : v;
}

}  // namespace

// The query-split area sits behind the fused area, at the tail of nw_fwd_workspace_bytes (capi.hip sizes it).
int split_queries_into_workspace(const float* q, void* workspace, size_t workspace_bytes, int64_t B, int64_t N, int64_t d,
                                 int64_t C, float** rows, float** scale, float** norm2, hipStream_t st) {
    const size_t total = nw_fwd_workspace_bytes(B, N, d, C);
    if (!workspace || workspace_bytes < total) return NW_ERR_WORKSPACE;
    const size_t a = al256((size_t)B * (size_t)d * sizeof(float)), b = al256((size_t)B * sizeof(float));
    char* base = static_cast<char*>(workspace) + (total - a - 3 * b);  // the last b bytes: the log-sum-exp slot (capi.hip)
    *rows = reinterpret_cast<float*>(base);
    *scale = reinterpret_cast<float*>(base + a);
    *norm2 = reinterpret_cast<float*>(base + a + b);
    return launch_split_rows(q, *rows, *scale, *norm2, B, d, st);
}

int launch_run_tables(const FusedWs& ws, const int64_t* sy, int N, int C, int n_stiles, int BS, hipStream_t st) {
    if (BS > 192) return NW_ERR_UNSUPPORTED;  // three rows per lane
    hipLaunchKernelGGL(nw_run_tables_kernel, dim3(n_stiles), dim3(64), 0, st, sy, N, C, BS, ws.runid, ws.nrun, ws.lab, ws.bnd);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

// A resident bank's run tables, built once (nw_bank_tables_build) instead of once per forward.  They depend on the
// labels, N, C and the tile height only; the persistent kernel's tile is 128 supports.  The caller passes them with the
// call (nw_fwd_opts.tables): they must be the tables of the very label array, N and C of that call.
constexpr int BANK_BS = 128;

size_t bank_tables_layout(int64_t n_stiles, char* base, FusedWs* ws) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += al256(bytes);
        return p;
    };
    int* nrun = reinterpret_cast<int*>(take((size_t)n_stiles * 4));
    int* lab = reinterpret_cast<int*>(take((size_t)n_stiles * BANK_BS * 4));
    int* runid = reinterpret_cast<int*>(take(((size_t)n_stiles * BANK_BS + 64) * 4));
    int* bnd = reinterpret_cast<int*>(take((size_t)n_stiles * 2 * 4));
    if (ws) {
        ws->nrun = nrun;
        ws->lab = lab;
        ws->runid = runid;
        ws->bnd = bnd;
    }
    return off;
}

bool bank_tables_take(const int64_t* sy, int N, int C, int n_stiles, int BS, FusedWs* ws) {
    (void)C;   // (the tables do not depend on the class count: labels >= C are the caller's check, as in nw_bank_tables_build)
    const FwdOpts& o = fwd_opts();
    if (!o.tables || BS != BANK_BS) return false;
    if (o.tables_sy != sy || o.tables_N != (int64_t)N) return false;   // tables of another label array: build our own
    if (o.tables_bytes < bank_tables_layout(n_stiles, nullptr, nullptr)) return false;
    bank_tables_layout(n_stiles, const_cast<char*>(o.tables), ws);
    return true;
}

int launch_merge_runs(const FusedWs& ws, float* out, float* lse, float* m, float* den, float* num,
                      int B, int C, int n_stiles, int BS, hipStream_t st) {
    // LDS: reduction scratch + per-class tables (count, first / last tile, MENT entries) + res[class][query]
    auto lds_bytes = [&](int mq, bool tables) {
        return ((size_t)2 * (MTHREADS / 64) * mq + 2 * mq + (tables ? (size_t)C * (3 + MENT + mq + 1) : 0)) * sizeof(float);
    };
    const size_t cap = 150 * 1024;
    // 32 queries x 16 tile lanes per workgroup when a query has few tiles (a shard of the bank), else 16 x 32;
    // below 512 queries one workgroup per query (T: 256 workgroups, one per CU)
    int mq = 1;
    const int force_mq = knob(KNOB_MERGE_MQ) == KNOB_UNSET ? 0 : knob(KNOB_MERGE_MQ);   // timing experiments
    if (force_mq == 1 || force_mq == 16 || force_mq == 32) {
        mq = force_mq;
        if (lds_bytes(mq, true) > cap) mq = 1;
    } else if (B >= 512 && !env_flag("NW_MERGE_PER_QUERY")) {
        mq = n_stiles >= 128 ? 16 : 32;
        if (lds_bytes(mq, true) > cap) mq = 16;
        if (lds_bytes(mq, true) > cap) mq = 1;
    }
    const bool tables = lds_bytes(mq, true) <= cap;
    const size_t lds = lds_bytes(mq, tables);
    const int grid = (B + mq - 1) / mq;
#define NW_MERGE(P_, Q_, T_)                                                                                       \
    hipLaunchKernelGGL((nw_merge_runs_kernel<P_, Q_, T_>), dim3(grid), dim3(MTHREADS), lds, st, ws.m, ws.den, ws.nrun, \
                       ws.lab, ws.num, out, lse, m, den, num, B, C, n_stiles, BS, ws.ctab)
    if (!tables) {
        // more classes than LDS holds tables for (C > ~4200): the tables go to the workspace, built once per launch
        // (measured at C = 5000: 4.9 ms per call at T when every class scanned every tile, 35 ms at B = 4096, N = 50000)
        if (ws.ctab && !env_flag("NW_MERGE_NO_GLOBAL_TABLES")) {
            hipLaunchKernelGGL(nw_class_tables_kernel, dim3(1), dim3(1024), 0, st, ws.nrun, ws.lab, n_stiles, BS, C, ws.ctab);
            if (out) NW_MERGE(false, 1, 2); else NW_MERGE(true, 1, 2);
        } else {
            if (out) NW_MERGE(false, 1, 0); else NW_MERGE(true, 1, 0);
        }
    } else if (mq == 1) {
        if (out) NW_MERGE(false, 1, 1); else NW_MERGE(true, 1, 1);
    } else if (mq == 16) {
        if (out) NW_MERGE(false, 16, 1); else NW_MERGE(true, 16, 1);
    } else {
        if (out) NW_MERGE(false, 32, 1); else NW_MERGE(true, 32, 1);
    }
#undef NW_MERGE
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

size_t fused_workspace_bytes(int64_t B, int64_t N, int64_t d, int64_t C) {
    size_t need = 0;
    for (int f16 = 0; f16 < 2; ++f16) {  // either operand form may be chosen at launch
        const int rs = pick_rs(B, N, d, f16 != 0);
        const int64_t n_stiles = (N + 16 * rs - 1) / (16 * rs);
        const size_t n = fused_layout(B, n_stiles, 16 * rs, nullptr, nullptr, C);
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
                 const float* s_scale, const float* ls, float* out,
                 float* scores, float* lse, float* m, float* den, float* num, void* workspace,
                 size_t workspace_bytes, int64_t B, int64_t N, int64_t d, int64_t C, int kind,
                 hipStream_t st) {
#define NW_KIND_CASE(K) \
    case K: return launch_fused_kind<K>(q, s, sy, s_norm2, s_scale, ls, out, scores, lse, m, den, num, workspace, workspace_bytes, (int)B, (int)N, (int)d, (int)C, st)
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

extern "C" size_t nw_bank_tables_bytes(int64_t N) {
    if (N <= 0 || N > 0x7fffffffLL) return 0;
    return nw::bank_tables_layout((N + nw::BANK_BS - 1) / nw::BANK_BS, nullptr, nullptr);
}

extern "C" int nw_bank_tables_build(const int64_t* sy, int64_t N, int64_t C, void* tables, size_t tables_bytes,
                                    void* stream) {
    using namespace nw;
    if (N < 0 || C < 0 || N > 0x7fffffffLL || C > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    if (N == 0) return NW_OK;
    if (!sy || !tables || (reinterpret_cast<uintptr_t>(tables) & 15)) return NW_ERR_INVALID_ARG;
    const int64_t n_stiles = (N + BANK_BS - 1) / BANK_BS;
    FusedWs ws;
    if (tables_bytes < bank_tables_layout(n_stiles, static_cast<char*>(tables), &ws)) return NW_ERR_WORKSPACE;
    return launch_run_tables(ws, sy, (int)N, (int)C, (int)n_stiles, BANK_BS, static_cast<hipStream_t>(stream));
}

