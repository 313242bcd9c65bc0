// fused_impl.h -- nw_fused_kernel and its per-score-kind launcher (gfx950 / MI355X only).
// Included by fused_k*.hip, one translation unit per score kind so that the build parallelises.
#pragma once
#include <type_traits>
#include "tile_core.h"
#include "tile_dma.h"
#include "tile_f16.h"

namespace nw {

constexpr int RUN_CAP = 192;  // >= 16*RS for the largest RS

struct FusedWs {  // views into the caller's workspace
    float* m;     // [n_stiles][B]
    float* den;   // [n_stiles][B]
    int* nrun;    // [n_stiles]
    int* lab;     // [n_stiles][BS]      label of run r (-1: padding / out-of-range label)
    float* num;   // [n_stiles][BS][B]   run sums, rows >= nrun[st] never touched
    int* runid;   // [n_stiles][BS] (+64)  run of every support row inside its tile (persistent kernel only)
    int* bnd;     // [n_stiles][2]  first tile rows of runs 1 and 2 (BS when there is no such run)
    int* ctab;    // [C][3 + MENT]  the run merge's class tables when they do not fit in LDS (nw_class_tables_kernel)
};
size_t fused_layout(int64_t B, int64_t n_stiles, int BS, char* base, FusedWs* ws, int64_t C = 0);
int launch_run_tables(const FusedWs& ws, const int64_t* sy, int N, int C, int n_stiles, int BS, hipStream_t st);
bool bank_tables_take(const int64_t* sy, int N, int C, int n_stiles, int BS, FusedWs* ws);   // the caller's cached tables, if named for this call
int launch_merge_runs(const FusedWs& ws, float* out, float* lse, float* m, float* den, float* num,
                      int B, int C, int n_stiles, int BS, hipStream_t st);
// split form of the query batch in the tail of the forward workspace (fused.hip)
int split_queries_into_workspace(const float* q, void* workspace, size_t workspace_bytes, int64_t B, int64_t N, int64_t d,
                                 int64_t C, float** rows, float** scale, float** norm2, hipStream_t st);
int device_cu_count();
bool env_flag(const char* name);
int tile_timer_start(hipStream_t st);          // diagnostics (nw_debug_tile_timing): -1 when disabled
void tile_timer_stop(int slot, hipStream_t st);
int persistent_qgroup();  // query tiles kept L2-resident per XCD by the persistent kernel (NW_QG)
int persistent_variant();  // NW_PVAR = 0: 64-query tiles, one workgroup per CU; 1: two per CU; 2: 128-query tiles; unset: -1

namespace {

// MODE_REG : register-staged loaders (tile_core.h), any d % 4 == 0; loaders compute both norms
// MODE_DMA : LDS-DMA loaders (tile_dma.h), d % 32 == 0; consumers compute both norms
// MODE_DMA_SN : LDS-DMA loaders, support norms supplied by the caller (cached bank)
// MODE_F16 : LDS-DMA loaders, split-fp16 operands on the fp16 matrix cores (tile_f16.h): q and s are
//            SPLIT rows, norms and row scales of both are supplied
// MODE_F16Q: as MODE_F16, but q holds the caller's RAW fp32 rows: the consumer waves compute the row scales and
//            norms in their prologue and split their query fragments in registers (tile_f16.h, QRAW) -- the
//            one-workgroup-per-tile kernel of small grids (T) runs without a query-split launch in front
//   (Measured and dropped, T shape: the query fragments RESIDENT in the consumer waves -- high halves in 64 VGPRs, low
//    halves parked in LDS, only support rows in the stage ring, one set of support fragments refilled block by block.
//    29 % fewer bytes through the loop's L2 -> LDS stream, but getting the rows into operand shape cost 11.5 k cycles
//    per workgroup (fragment-shaped global loads, or sixteen 8 KB DMA steps each paying a third of the DMA latency)
//    and the single-buffered loop ran 750 cycles per stage against 530: 18.5-19.4 us against 16.4.)
enum { MODE_REG = 0, MODE_DMA = 1, MODE_DMA_SN = 2, MODE_F16 = 3, MODE_F16Q = 4 };
constexpr bool mode_is_f16(int m) { return m == MODE_F16 || m == MODE_F16Q; }

// Runs of equal consecutive labels inside one support tile, by ONE wave (3 rows per lane): fills
// runid[t] (run of tile row t), runlab[run] (its class, -1 = padding / out-of-range label), nrun_s[0] =
// number of runs and nrun_s[1], nrun_s[2] = first tile rows of runs 1 and 2 (BS when there is none).
// `lab` = this lane's three labels (tile rows 3*lane .. 3*lane+2), already mapped to -1 when invalid.
template <int BS>
__device__ __forceinline__ void run_scan_wave(const int (&lab)[3], int lane, int* runid, int* runlab, int* nrun_s) {
    int flag[3];
    if (lane < 2) nrun_s[1 + lane] = BS;  // overwritten below by the lane that starts run 1 / run 2 (same wave: in order)
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
    int id = incl - (flag[0] + flag[1] + flag[2]) - 1;  // run id before this lane's rows
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = 3 * lane + u;
        id += flag[u];
        if (t < BS) {
            runid[t] = id;
            if (flag[u] && (id == 1 || id == 2)) nrun_s[id] = t;
            if (flag[u]) runlab[id] = lab[u];
            if (t == BS - 1) *nrun_s = id + 1;
        }
    }
}
template <int BS>
__device__ __forceinline__ void load_tile_labels(const int64_t* __restrict__ sy, int s0, int N, int C, int lane, int (&lab)[3]) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = 3 * lane + u, j = s0 + t;
        int64_t y = -1;
        if (t < BS && j < N) y = sy[j];
        lab[u] = ((uint64_t)y < (uint64_t)C) ? (int)y : -1;
    }
}

#ifdef NW_DIAG_FUSED   // diagnostic build only (tools/bench_fused.hip): phase stamps go to `scores`
#define NW_FSTAMP(k) if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(scores)[8 * blockIdx.x + (k)] = __builtin_amdgcn_s_memtime()
#else
#define NW_FSTAMP(k)
#endif
// The epilogue of one tile: scores -> tile-local softmax statistics -> run sums -> workspace.
// Called by all threads of the workgroup (loader waves only take part in the run-table copy).
template <int RS, int KIND, bool WRITE_SCORES, int MODE>
__device__ __forceinline__ void fused_epilogue(
    f32x4 (&acc)[RS], const float* qn2, const float* sn2, const float* ssc, const int* runid,
    const int* runlab, const int* nrun_s, const float* qsc_s,
    const float* __restrict__ logit_scale, float* __restrict__ scores, float* __restrict__ ws_m,
    float* __restrict__ ws_den, int* __restrict__ ws_nrun, int* __restrict__ ws_lab,
    float* __restrict__ ws_num, int B, int N, int q0, int s0, int qt, int st) {
    constexpr int BS = 16 * RS;
    constexpr bool NEED_NORM = (KIND != NW_SCORE_DOT);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, g = lane >> 4;
    const bool consumer = wave < NCONS;  // waves 4-7 (loaders) hold no accumulators
    // The softmax part works in BASE-2 units u = score * log2(e): e^(s-m) = 2^(u-mu) is then one
    // subtract and one v_exp_f32, and for the Euclidean kernel the constant is folded into the
    // squared distance (u = -sqrt(log2(e)^2 * d2)), so an element costs fma, fma, max, sqrt, sub,
    // exp2 + its share of the reductions -- in an fp32-MFMA kernel every VALU instruction is
    // matrix-pipe time.  Tile statistics (ws_m) are kept in base-2 units; the merge converts.
    constexpr float L2E = 1.44269504088896340736f, LN2 = 0.693147180559945309417f;
    float scale = 1.f;
    if (KIND == NW_SCORE_CLIP) scale = expf(*logit_scale);
    const int qrow = 16 * (wave & 3) + i;
    const int b = q0 + qrow;
    const float qn = NEED_NORM ? qn2[qrow] : 0.f;
    const float qsc = mode_is_f16(MODE) ? qsc_s[qrow] : 1.f;  // 2^-e of this lane's query row (LDS header)
    const bool partial_tile = s0 + BS > N;  // only the last support tile has rows past the bank

    float sc[RS][4];
    float mloc = -INFINITY;
    if (consumer) {
        // per-row score factors (nw_internal.h, ScoreFactors): x = acc * K * Cq + (Base + Bq)
        using SF = ScoreFactors<KIND>;
        float Cq, Bq;
        SF::query(qn, qsc, scale, Cq, Bq);
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            float4 n2 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (NEED_NORM) n2 = *reinterpret_cast<const float4*>(sn2 + 16 * r + 4 * g);
            if (mode_is_f16(MODE)) s4 = *reinterpret_cast<const float4*>(ssc + 16 * r + 4 * g);  // 2^-e of the support rows
            const float nn[4] = {n2.x, n2.y, n2.z, n2.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float K, Base;
                SF::support(nn[e], ss[e], K, Base);
                sc[r][e] = SF::finish(__builtin_fmaf(acc[r][e], K * Cq, Base + Bq));
            }
        }
        if (partial_tile) {
#pragma unroll
            for (int r = 0; r < RS; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (s0 + 16 * r + 4 * g + e >= N) sc[r][e] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < RS; ++r) mloc = fmaxf(mloc, fmaxf(fmaxf(sc[r][0], sc[r][1]), fmaxf(sc[r][2], sc[r][3])));
        if (WRITE_SCORES && b < B) {  // natural units for the caller (backward, neighbour search)
            float* orow = scores + (size_t)b * N;
            const bool vec_ok = (N & 3) == 0;
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                const int j = s0 + 16 * r + 4 * g;
                if (j >= N) continue;
                if (vec_ok) {
                    *reinterpret_cast<float4*>(orow + j) =
                        make_float4(sc[r][0] * LN2, sc[r][1] * LN2, sc[r][2] * LN2, sc[r][3] * LN2);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j + e < N) orow[j + e] = sc[r][e] * LN2;
                }
            }
        }
        // tile-local max over the wave's 16 query columns: lanes i, i+16, i+32, i+48 hold one query
        mloc = group4_max(mloc);
    }
    NW_FSTAMP(3);

    // ---- 2^(u - mu) and its sums over the runs of equal labels, on the matrix cores:
    //   P[run][query] = sum_t [runid_t == run] * E[t][query]
    // E is already laid out as an MFMA B operand (lane (i,g) holds E[16r+4g+e][query i]: for fixed
    // (r,e) the four lane groups are the four k-slots of one 16x16x4 MFMA), the indicator is the A
    // operand (lane (i,g) supplies [runid[16r+4g+e] == run_base + i]), so 4*RS MFMAs per 16 runs give
    // every lane its four (run, query) sums: no LDS atomics, no divergence, bit-reproducible.
    if (consumer) {
        float dloc = 0.f;
#pragma unroll
        for (int r = 0; r < RS; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[r][e] = __builtin_amdgcn_exp2f(sc[r][e] - mloc);  // 2^-inf = 0 for padded rows
        // Run sums.  A run is a RANGE of tile rows: with the first rows b1, b2 of runs 1 and 2 the
        // membership of row t is a clamped difference, [t < b] = clamp(b - t, 0, 1) -- for up to three
        // runs (a class-sorted bank has one or two per tile) 2-7 VALU ops per element instead of a
        // dependent chain of 4*RS fp32 MFMAs; more runs go through the indicator MFMAs.
        const int nrun = nrun_s[0];
        if (nrun <= 3) {
            float S0[2] = {0.f, 0.f}, S1[2] = {0.f, 0.f}, S2[2] = {0.f, 0.f};
            if (nrun == 1) {
#pragma unroll
                for (int r = 0; r < RS; ++r) S0[r & 1] += (sc[r][0] + sc[r][1]) + (sc[r][2] + sc[r][3]);
            } else if (nrun == 2) {
                const float L1 = (float)(nrun_s[1] - 4 * g), M1 = 1.f - L1;
#pragma unroll
                for (int r = 0; r < RS; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float c = (float)(16 * r + e), ev = sc[r][e];
                        S0[e & 1] = __builtin_fmaf(__builtin_amdgcn_fmed3f(L1 - c, 0.f, 1.f), ev, S0[e & 1]);  // [t <  b1]
                        S1[e & 1] = __builtin_fmaf(__builtin_amdgcn_fmed3f(c + M1, 0.f, 1.f), ev, S1[e & 1]);  // [t >= b1]
                    }
            } else {
                const float L1 = (float)(nrun_s[1] - 4 * g);
                const float M2 = 1.f - (float)(nrun_s[2] - 4 * g);
#pragma unroll
                for (int r = 0; r < RS; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float c = (float)(16 * r + e), ev = sc[r][e];
                        const float w1 = __builtin_amdgcn_fmed3f(L1 - c, 0.f, 1.f);  // [t <  b1]
                        const float u2 = __builtin_amdgcn_fmed3f(c + M2, 0.f, 1.f);  // [t >= b2]
                        S0[e & 1] = __builtin_fmaf(w1, ev, S0[e & 1]);
                        S2[e & 1] = __builtin_fmaf(u2, ev, S2[e & 1]);
                        S1[e & 1] = __builtin_fmaf((1.f - w1) - u2, ev, S1[e & 1]);  // exact 0 / 1
                    }
            }
            const float s0v = group4_sum(S0[0] + S0[1]);
            float s1v = 0.f, s2v = 0.f;
            if (nrun >= 2) s1v = group4_sum(S1[0] + S1[1]);
            if (nrun == 3) s2v = group4_sum(S2[0] + S2[1]);
            dloc = (s0v + s1v) + s2v;
            if (g == 0 && b < B) {
                ws_num[((size_t)st * BS) * B + b] = s0v;
                if (nrun >= 2) ws_num[((size_t)st * BS + 1) * B + b] = s1v;
                if (nrun == 3) ws_num[((size_t)st * BS + 2) * B + b] = s2v;
            }
        } else {
            float dl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RS; ++r) dl[r & 3] += (sc[r][0] + sc[r][1]) + (sc[r][2] + sc[r][3]);
            dloc = group4_sum((dl[0] + dl[1]) + (dl[2] + dl[3]));
            for (int run_base = 0; run_base < nrun; run_base += 16) {
                // four independent accumulation chains (a dependent fp32 MFMA waits 40 cycles, an independent
                // one issues every 32), added at the end in a fixed order
                f32x4 P0 = {0.f, 0.f, 0.f, 0.f}, P1 = P0, P2 = P0, P3 = P0;
                const int want = run_base + i;
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    const int4 rid = *reinterpret_cast<const int4*>(runid + 16 * r + 4 * g);
                    P0 = __builtin_amdgcn_mfma_f32_16x16x4f32(rid.x == want ? 1.f : 0.f, sc[r][0], P0, 0, 0, 0);
                    P1 = __builtin_amdgcn_mfma_f32_16x16x4f32(rid.y == want ? 1.f : 0.f, sc[r][1], P1, 0, 0, 0);
                    P2 = __builtin_amdgcn_mfma_f32_16x16x4f32(rid.z == want ? 1.f : 0.f, sc[r][2], P2, 0, 0, 0);
                    P3 = __builtin_amdgcn_mfma_f32_16x16x4f32(rid.w == want ? 1.f : 0.f, sc[r][3], P3, 0, 0, 0);
                }
                const f32x4 P = (P0 + P1) + (P2 + P3);
                // P[j] = sum of run (run_base + 4g + j) for query column i
                if (b < B) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int run = run_base + 4 * g + j;
                        if (run < nrun) ws_num[((size_t)st * BS + run) * B + b] = P[j];
                    }
                }
            }
        }
        NW_FSTAMP(4);
        if (g == 0 && b < B) {
            ws_m[(size_t)st * B + b] = mloc;
            ws_den[(size_t)st * B + b] = dloc;
        }
    }
    NW_FSTAMP(5);
    if (qt == 0) {  // run table is a property of the support tile: written once per tile
        const int nrun = *nrun_s;
        if (tid == 0) ws_nrun[st] = nrun;
        for (int x = tid; x < nrun; x += TILE_THREADS) ws_lab[(size_t)st * BS + x] = runlab[x];
    }
}


template <int RS, int KIND, bool WRITE_SCORES, int MODE>
__global__ __launch_bounds__(TILE_THREADS, (RS <= 5 ? 4 : 2)) void nw_fused_kernel(
    const float* __restrict__ q, const float* __restrict__ s, const int64_t* __restrict__ sy,
    const float* __restrict__ s_norm2, const float* __restrict__ s_scale, const float* __restrict__ q_norm2,
    const float* __restrict__ q_scale, const float* __restrict__ logit_scale,
    float* __restrict__ scores, float* __restrict__ ws_m,
    float* __restrict__ ws_den, int* __restrict__ ws_nrun, int* __restrict__ ws_lab,
    float* __restrict__ ws_num, int B, int N, int d, int C, int n_stiles, int n_qtiles) {
    using Cfg = TileCfg<RS>;
    constexpr int BS = Cfg::BS;
    constexpr bool NEED_NORM = (KIND != NW_SCORE_DOT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // header: qn2[64] | qsc[64] | sn2[RUN_CAP] | ssc[RUN_CAP] | runid[RUN_CAP] | runlab[RUN_CAP] | nrun ; then the ring
    float* qn2 = reinterpret_cast<float*>(smem);
    float* qsc_s = qn2 + 64;      // MODE_F16: per-query row scale 2^-e
    float* sn2 = qsc_s + 64;
    float* ssc = sn2 + RUN_CAP;   // MODE_F16: per-support row scale 2^-e
    int* runid = reinterpret_cast<int*>(ssc + RUN_CAP);
    int* runlab = runid + RUN_CAP;
    int* nrun_s = runlab + RUN_CAP;
    constexpr int HDR = (128 + 4 * RUN_CAP + 4) * 4;
    static_assert(HDR % 16 == 0, "stage buffers must stay 16-byte aligned");
    float4* stage = reinterpret_cast<float4*>(smem + HDR);

    int qt, st;
    if (!decode_block(n_stiles, n_qtiles, qt, st)) return;
    const int q0 = qt * BQ, s0 = st * BS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, g = lane >> 4;
    NW_FSTAMP(0);

    // cached support norms: fetched now, long before the epilogue needs them (the DMA loop never
    // touches sn2 in this mode)
    if ((MODE == MODE_DMA_SN || mode_is_f16(MODE)) && NEED_NORM) {
        for (int t = tid; t < BS; t += TILE_THREADS) sn2[t] = s_norm2[min(s0 + t, N - 1)];
    }
    if (mode_is_f16(MODE)) {
        for (int t = tid; t < BS; t += TILE_THREADS) ssc[t] = s_scale[min(s0 + t, N - 1)];
    }
    if (MODE == MODE_F16) {  // MODE_F16Q: the consumer waves fill qn2 / qsc_s themselves
        if (NEED_NORM)
            for (int t = tid; t < BQ; t += TILE_THREADS) qn2[t] = q_norm2[min(q0 + t, B - 1)];
        for (int t = tid; t < BQ; t += TILE_THREADS) qsc_s[t] = q_scale[min(q0 + t, B - 1)];
    }
    // ---- runs of equal consecutive labels inside this support tile (one wave)
    if (wave == 0) {
        int lab[3];
        load_tile_labels<BS>(sy, s0, N, C, lane, lab);
        run_scan_wave<BS>(lab, lane, runid, runlab, nrun_s);
    }

    NW_FSTAMP(1);
    // the K walk of every support tile starts at a different chunk (see tile_core.h: spreads the
    // simultaneous requests of all workgroups over the memory channels); the workgroups that share
    // a support tile keep the same order so that they still hit each other's lines in L2
    const int nk = (d + BK - 1) / BK;
    const int rot = st % nk;
    f32x4 acc[RS];
    if (mode_is_f16(MODE)) {
        tile_dots_f16x2<RS, MODE == MODE_F16Q>(q, s, B, N, d, q0, s0, stage, acc, rot, qn2, qsc_s);
        __syncthreads();  // header tables written at kernel start are visible; the ring is dead
    } else if (MODE == MODE_REG) {
        tile_dots<RS, NEED_NORM>(q, s, B, N, d, q0, s0, stage, qn2, sn2, acc, rot);
    } else {
        tile_dots_dma<RS, NEED_NORM, NEED_NORM && MODE == MODE_DMA>(q, s, B, N, d, q0, s0, stage, qn2, sn2, acc, rot);
    }
    // (both end behind a barrier: the run tables above and the norms are visible, and the stage
    //  buffers are dead from here on)
    if (MODE != MODE_DMA_SN && !mode_is_f16(MODE) && NEED_NORM && s_norm2 != nullptr) {  // cached norms win over computed ones
        for (int t = tid; t < BS; t += TILE_THREADS) sn2[t] = s_norm2[min(s0 + t, N - 1)];
        __syncthreads();
    }

    NW_FSTAMP(2);
    fused_epilogue<RS, KIND, WRITE_SCORES, MODE>(acc, qn2, sn2, ssc, runid, runlab, nrun_s, qsc_s, logit_scale,
                                                 scores, ws_m, ws_den, ws_nrun, ws_lab, ws_num, B, N, q0, s0, qt, st);
    NW_FSTAMP(6);
}


constexpr size_t FUSED_HDR = (128 + 4 * RUN_CAP + 4) * 4;

template <int RS, int KIND>
int launch_f16p(const float* q, const float* s, const int64_t* sy, const float* s_norm2, const float* s_scale,
                const float* q_norm2, const float* q_scale, const float* ls, const FusedWs& ws, int B, int N,
                int d, int C, int n_stiles, int n_qtiles, hipStream_t st);

// q: the caller's RAW fp32 queries.  s_scale != nullptr: s holds the SPLIT rows of a prepared bank (d % 32 == 0,
// s_norm2 given).  The persistent kernel and the score-writing variant take split queries: the split launch
// (nw_split_rows_kernel into the tail of the workspace) happens here, only for them.
template <int RS, int KIND>
int launch_fused_rs(const float* q, const float* s, const int64_t* sy, const float* s_norm2,
                    const float* s_scale,
                    const float* ls, float* out, float* scores, float* lse, float* m, float* den,
                    float* num, void* workspace, size_t workspace_bytes, int B, int N, int d, int C,
                    hipStream_t st) {
    const float *q_norm2 = nullptr, *q_scale = nullptr;
    constexpr int BS = 16 * RS;
    const int n_stiles = (N + BS - 1) / BS;
    const int n_qtiles = (B + BQ - 1) / BQ;
    FusedWs ws;
    const size_t need = fused_layout(B, n_stiles, BS, static_cast<char*>(workspace), &ws, C);
    if (!workspace || workspace_bytes < need) return NW_ERR_WORKSPACE;
    const int grid = padded_grid(n_stiles, n_qtiles);
    // RS = 5 exists for the LDS-DMA modes only (two workgroups per CU); the register-staged loaders
    // need an even split of the tile rows
    const bool dma = (d % BK) == 0 && (uint64_t)(BQ + BS) * 2 * d * 4 < 0xffffffffull;  // per-lane offsets are tile-relative
    if (!dma && (RS & 1)) return NW_ERR_UNSUPPORTED;
    const size_t lds_reg = FUSED_HDR + TileCfg<RS>::STAGE_BYTES, lds_dma = FUSED_HDR + DmaCfg<RS>::STAGE_BYTES;
#define NW_LAUNCH(WS_, MODE_, LDS_)                                                                      \
    hipLaunchKernelGGL((nw_fused_kernel<RS, KIND, WS_, MODE_>), dim3(grid), dim3(TILE_THREADS), LDS_, st, \
                       q, s, sy, s_norm2, s_scale, q_norm2, q_scale, ls, scores, ws.m, ws.den, ws.nrun, ws.lab, \
                       ws.num, B, N, d, C,                                                                  \
                       n_stiles, n_qtiles)
    // many tiles per CU on split operands: the persistent kernel (fused_f16p.h).  RS = 8 is the tallest
    // tile whose build stays under 256 VGPRs.
    const bool persistent = s_scale && !scores && RS > 5 && (RS == 8 || env_flag("NW_PERSISTENT_ANY_RS")) &&
                            grid >= 4 * device_cu_count() && d >= 3 * BK && !env_flag("NW_NO_PERSISTENT");
    if (persistent) {  // runs of equal labels per support tile: once per launch (ws.runid / nrun / lab / bnd)
        if (!bank_tables_take(sy, N, C, n_stiles, 16 * RS, &ws)) {
            const int rc = launch_run_tables(ws, sy, N, C, n_stiles, 16 * RS, st);
            if (rc != NW_OK) return rc;
        }
    }
    if (s_scale) {  // split-fp16 operands (the caller has checked d % 32 == 0 and supplied the bank's norms)
        if (!dma || !s_norm2) return NW_ERR_INVALID_ARG;
        // Raw queries (MODE_F16Q) cost every workgroup a pass over its 64 query rows and the split in its loop
        // (~5.4 k cycles at T); the split launch costs ~4.4 us + a kernel boundary once.  Measured per forward
        // (N = 10000, d = 512; raw / split launch): B = 256 20.2 / 21.8 us, 512 40.6 / 40.1, 768 52.7 / 50.4, 1000
        // 62.4 / 62.3: raw up to 1.5 workgroups per CU.  NW_SPLIT_QUERIES=1 / 0 forces either.
        const int force_split = knob(KNOB_SPLIT_QUERIES) == KNOB_UNSET ? -1 : knob(KNOB_SPLIT_QUERIES);
        const bool raw_ok = force_split == 0 || (force_split < 0 && 2 * grid <= 3 * device_cu_count());
        if (persistent || !raw_ok) {
            float *qr, *qsc, *qn;
            const int rc = split_queries_into_workspace(q, workspace, workspace_bytes, B, N, d, C, &qr, &qsc, &qn, st);
            if (rc != NW_OK) return rc;
            q = qr;
            q_scale = qsc;
            q_norm2 = qn;
        }
    }
    const int timer_slot = tile_timer_start(st);
    if (s_scale) {
        if (!q_scale) {
            if (scores) NW_LAUNCH(true, MODE_F16Q, lds_dma); else NW_LAUNCH(false, MODE_F16Q, lds_dma);
        } else if (scores) {
            NW_LAUNCH(true, MODE_F16, lds_dma);
        } else if (persistent) {
            const int rc = launch_f16p<RS, KIND>(q, s, sy, s_norm2, s_scale, q_norm2, q_scale, ls, ws, B, N, d, C,
                                                 n_stiles, n_qtiles, st);
            if (rc != NW_OK) return rc;
        } else {
            NW_LAUNCH(false, MODE_F16, lds_dma);
        }
    } else if (dma && s_norm2 && KIND != NW_SCORE_DOT) {
        if (scores) NW_LAUNCH(true, MODE_DMA_SN, lds_dma); else NW_LAUNCH(false, MODE_DMA_SN, lds_dma);
    } else if (dma) {
        if (scores) NW_LAUNCH(true, MODE_DMA, lds_dma); else NW_LAUNCH(false, MODE_DMA, lds_dma);
    } else {
        if (scores) NW_LAUNCH(true, MODE_REG, lds_reg); else NW_LAUNCH(false, MODE_REG, lds_reg);
    }
#undef NW_LAUNCH
    tile_timer_stop(timer_slot, st);
    NW_CHECK_LAUNCH();
    return launch_merge_runs(ws, out, lse, m, den, num, B, C, n_stiles, BS, st);
}

}  // namespace

template <int KIND>
int launch_fused_kind(const float* q, const float* s, const int64_t* sy, const float* s_norm2,
                      const float* s_scale,
                      const float* ls, float* out, float* scores, float* lse, float* m, float* den,
                      float* num, void* workspace, size_t wsb, int B, int N, int d, int C, hipStream_t st) {
#define NW_RS_CASE(R) \
    case R: return launch_fused_rs<R, KIND>(q, s, sy, s_norm2, s_scale, ls, out, scores, lse, m, den, num, workspace, wsb, B, N, d, C, st)
    switch (pick_rs(B, N, d, s_scale != nullptr)) {
        NW_RS_CASE(2);
        NW_RS_CASE(4);
        NW_RS_CASE(5);
        NW_RS_CASE(6);
        NW_RS_CASE(8);
        NW_RS_CASE(10);
        default: return launch_fused_rs<12, KIND>(q, s, sy, s_norm2, s_scale, ls, out, scores, lse, m, den, num, workspace, wsb, B, N, d, C, st);
    }
#undef NW_RS_CASE
}

}  // namespace nw
#include "fused_f16p.h"
#ifdef NW_WITH_P8   // tools/bench_fused.hip only: the eight-multiplying-wave experiment (tools/experiments/fused_f16p8.h)
#include "fused_f16p8.h"
#endif
namespace nw {
namespace {
template <int RS, int KIND>
int launch_f16p(const float* q, const float* s, const int64_t* sy, const float* s_norm2, const float* s_scale,
                const float* q_norm2, const float* q_scale, const float* ls, const FusedWs& ws, int B, int N,
                int d, int C, int n_stiles, int n_qtiles, hipStream_t st) {
    if constexpr (RS > 5) {
        (void)sy; (void)C;  // the run tables (launch_run_tables) are the caller's job
        int cus = device_cu_count() & ~7;  // the same number of workgroups on every XCD
        // nw_fwd_opts.persistent_wgs: fewer workgroups than CUs (a multiple of 8), to leave CUs to a concurrent RCCL kernel
        // of the sharded path (ShardedBank leaves one CU per XCD when there is more than one rank)
        const int wg_cap = fwd_opts().persistent_wgs & ~7;
        if (wg_cap >= 8 && wg_cap < cus) cus = wg_cap;
        // 0: 64-query tiles, one workgroup per CU; 1: two per CU; 2: 128-query tiles.  Measured at B = 2048,
        // N = 50000, d = 512 (tools/bench_fused.hip, same device): 387 / 353 / 337 us.  128-query tiles
        // unless their padding costs more than 15 % of the rows (then two 64-query workgroups per CU).
        int variant = persistent_variant();
        if (variant < 0) variant = ((B + 127) / 128 * 128 <= 1.15 * ((B + 63) / 64 * 64)) ? 2 : 1;
        (void)n_qtiles;
#ifdef NW_WITH_P8   // NW_PVAR=3 in tools/bench_fused.hip: 256-query tiles, eight multiplying waves (measured slower)
        if constexpr (RS == 8) {
            if (variant == 3) {
                static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(nw_fused_f16p8_kernel<KIND>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8::LDS_BYTES) == hipSuccess;
                (void)attr;
                int qgrp = persistent_qgroup() / 2;   // the same bytes of queries resident per XCD as with 128-query tiles
                if (qgrp < 1) qgrp = 1;
                hipLaunchKernelGGL((nw_fused_f16p8_kernel<KIND>), dim3(cus), dim3(TILE_THREADS), P8::LDS_BYTES, st, q, s, s_norm2,
                                   s_scale, q_norm2, q_scale, ls, ws.runid, ws.nrun, ws.bnd, ws.m, ws.den, ws.num, B, N, d, n_stiles,
                                   (B + P8::BQP - 1) / P8::BQP, qgrp);
                NW_CHECK_LAUNCH();
                return NW_OK;
            }
        }
#endif
        if (variant > 2) variant = 2;
#define NW_LAUNCH_P(TWO_, QB_, GRID_, NBUF_)                                                                      \
    do {                                                                                                          \
        using PC_ = PCfg<RS, QB_>;                                                                                \
        const size_t lds_ = PC_::HDR_BYTES + (size_t)(NBUF_) * PC_::TILE_F4 * 16;                                 \
        hipLaunchKernelGGL((nw_fused_f16p_kernel<RS, KIND, TWO_, QB_>), dim3(GRID_), dim3(TILE_THREADS), lds_, st, \
                           q, s, s_norm2, s_scale, q_norm2, q_scale, ls, ws.runid, ws.nrun, ws.bnd, ws.m, ws.den, ws.num,  \
                           B, N, d, n_stiles, (B + 64 * (QB_) - 1) / (64 * (QB_)), persistent_qgroup());          \
    } while (0)
        constexpr size_t lds_q2 = PCfg<RS, 2>::HDR_BYTES + (size_t)4 * PCfg<RS, 2>::TILE_F4 * 16;
        constexpr size_t lds_two = PCfg<RS, 1>::HDR_BYTES + (size_t)3 * PCfg<RS, 1>::TILE_F4 * 16;
        if (variant == 2 && lds_q2 <= 160 * 1024) {
            NW_LAUNCH_P(false, 2, cus, 4);
        } else if (variant == 1 && lds_two <= 80 * 1024) {
            NW_LAUNCH_P(true, 1, 2 * cus, 3);
        } else {
            NW_LAUNCH_P(false, 1, cus, 4);
        }
#undef NW_LAUNCH_P
        NW_CHECK_LAUNCH();
    }
    return NW_OK;
}
}  // namespace

#define NW_INSTANTIATE_FUSED_KIND(K)                                                                   \
    template int launch_fused_kind<K>(const float*, const float*, const int64_t*, const float*,        \
                                      const float*,                                                    \
                                      const float*, float*, float*, float*, float*, float*, float*,    \
                                      void*, size_t, int, int, int, int, hipStream_t);

}  // namespace nw
