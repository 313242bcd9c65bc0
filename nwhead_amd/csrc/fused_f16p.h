// fused_f16p.h -- persistent edition of the split-fp16 fused forward (gfx950 / MI355X only).
//
// One 512-thread workgroup per CU walks its share of the (query tile, support tile) grid instead of
// one workgroup per tile.  Measured on the one-workgroup-per-tile kernel (tools/bench_fused.hip,
// B=2048 N=50000): 21.9 k cycles of stamped work per tile but 27 k cycles per tile of wall time --
// every new workgroup pays its dispatch (LDS allocation, wave launch), the latency of its label/norm
// loads, the run scan and the DMA prologue before the first MFMA.  Here
//   * the loader waves keep ONE stage pipeline running across tile boundaries: while the consumers
//     run a tile's epilogue the first three stages of the next tile are already landing;
//   * the tile header (support norms, row scales, run ids of the support rows; norms and scales of
//     the query rows) is DMA'd into LDS by the loaders together with the first stage of the tile,
//     into one of three header buffers (tile index mod 3), so the consumers never wait for a global load;
//   * the runs of equal labels are found ONCE per launch by nw_run_tables_kernel (fused.hip) instead
//     of once per (query tile, support tile) workgroup;
//   * barriers: one per stage, executed by both roles, nothing else.
// Tile shape, LDS stage image and MFMA stream are those of nw_fused_kernel<RS, KIND, false, MODE_F16>.
// Requires d / 32 >= 3: the header of tile T+2 is issued 3 stages before tile T+1 ends, i.e. (three
// stages per tile) possibly while the consumers still run the epilogue of tile T -- hence three header
// buffers; with fewer stages per tile the issue cursor would run further ahead than that.
//
// What bounds it (tools/bench_f16_loop.hip, RS = 8, random operands): the main loop needs 24 MFMAs
// (384 cycles) and a 24.6 KB LDS fill per stage.  From an L2-resident source the loop runs 495
// cycles per stage, and the chip holds 1.50 GHz on it (2.37 GHz on zeros): 345 ns per stage = 388
// TFLOP/s-equivalent is the power-limited pace of this instruction mix, not 833.
#pragma once
#include "fused_impl.h"

namespace nw {
namespace {

#ifdef NW_DIAG_FUSED  // diagnostic build only (tools/bench_fused.hip): per-workgroup phase totals
__device__ unsigned long long nw_diag_p[8 * 1024];
__device__ unsigned long long nw_diag_rt[2 * 1024];   // s_memrealtime (100 MHz) at the first / last stamp of a workgroup
#define NW_PSTAMP(k)                                                                         \
    do {                                                                                     \
        unsigned long long now_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        diag_[k] += now_ - last_;                                                            \
        last_ = now_;                                                                        \
    } while (0)
#else
#define NW_PSTAMP(k)
#endif

// QB = query blocks (of 16 rows) per consumer wave: the tile is 64*QB queries x 16*RS supports.
template <int RS, int QB>
struct PCfg {
    static constexpr int BS = 16 * RS;
    static constexpr int BQP = 64 * QB;                 // query rows per tile
    static constexpr int N64 = (BS + 63) / 64;          // 64-row DMA pieces per support-side header array
    static constexpr int NH = N64 * 64;                 // entries per support-side array in LDS
    static constexpr int HDR_F = 3 * NH + 2 * BQP;      // sn2 | ssc | runid | qn2[BQP] | qsc[BQP]
    static constexpr int NP = 3 * N64 + 2 * QB;         // header pieces (256 B each)
    static constexpr int HPW = (NP + NLOAD - 1) / NLOAD;  // ... per loader wave
    static constexpr int TILE_F4 = (BQP + BS) * ROW_F4; // one stage: 128 B per row
    static constexpr int NT = (BQP + BS) / 8;           // stage DMA pieces (1 KB = 8 rows)
    static constexpr int NI = (NT + NLOAD - 1) / NLOAD; // ... per loader wave (waves lw < NT % NLOAD, or all)
    static constexpr int NI_LO = NT / NLOAD;
    static constexpr int NHB = 3;                       // header buffers
    static constexpr size_t HDR_BYTES = NHB * (size_t)HDR_F * 4;
    static_assert(HDR_BYTES % 16 == 0, "stage buffers must stay 16-byte aligned");
};

// Epilogue of one tile for the consumer waves: scores -> tile-local softmax statistics -> run sums.
// Same arithmetic as fused_epilogue<.., MODE_F16> (fused_impl.h); the header comes from LDS only.
// A wave owns QB blocks of 16 queries (rows 16*(QB*wave + j) + i of the tile) x all 16*RS supports.
// NCW = waves that own query rows (the header's query arrays hold 16 * QB * NCW entries).
template <int RS, int KIND, int QB, int NCW = NCONS>
__device__ __forceinline__ void epilogue_p(f32x4 (&acc)[QB][RS], const float* hdr, int nrun, int2 bnd,
                                           const float* __restrict__ logit_scale, float* __restrict__ ws_m,
                                           float* __restrict__ ws_den, float* __restrict__ ws_num, int B, int N,
                                           int q0, int s0, int st, int wave, int lane
#ifdef NW_DIAG_FUSED
                                           , unsigned long long (&diag_)[8], unsigned long long& last_
#endif
                                           ) {
    using P = PCfg<RS, QB>;
    constexpr int BS = P::BS;
    constexpr bool NEED_NORM = (KIND != NW_SCORE_DOT);
    constexpr float L2E = 1.44269504088896340736f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float* sn2 = hdr;
    const float* ssc = hdr + P::NH;
    const int* runid = reinterpret_cast<const int*>(hdr + 2 * P::NH);
    const float* qn2 = hdr + 3 * P::NH;
    const float* qsc_s = qn2 + 16 * QB * NCW;
    const int i = lane & 15, g = lane >> 4;
    float scale = 1.f;
    if (KIND == NW_SCORE_CLIP) scale = expf(*logit_scale);
    // every header vector this lane needs, in one burst of LDS reads; then the per-support-row score
    // factors (nw_internal.h, ScoreFactors), shared by the QB query blocks
    using SF = ScoreFactors<KIND>;
    float4 n4[RS], s4[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        n4[r] = NEED_NORM ? *reinterpret_cast<const float4*>(sn2 + 16 * r + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        s4[r] = *reinterpret_cast<const float4*>(ssc + 16 * r + 4 * g);
    }
    NW_PSTAMP(1);
    float4 K4[RS], B4[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        SF::support(n4[r].x, s4[r].x, K4[r].x, B4[r].x);
        SF::support(n4[r].y, s4[r].y, K4[r].y, B4[r].y);
        SF::support(n4[r].z, s4[r].z, K4[r].z, B4[r].z);
        SF::support(n4[r].w, s4[r].w, K4[r].w, B4[r].w);
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int qrow = 16 * (QB * wave + j) + i;
        const int b = q0 + qrow;
        float Cq, Bq;
        SF::query(NEED_NORM ? qn2[qrow] : 0.f, qsc_s[qrow], scale, Cq, Bq);
        const f32x2 cq = {Cq, Cq}, bq = {Bq, Bq};
        float sc[RS][4];
#pragma unroll
        for (int r = 0; r < RS; ++r) {  // x = acc * (K * Cq) + (Base + Bq): three packed fp32 ops per pair
            const f32x2 d01 = __builtin_elementwise_fma(f32x2{acc[j][r][0], acc[j][r][1]}, f32x2{K4[r].x, K4[r].y} * cq,
                                                        f32x2{B4[r].x, B4[r].y} + bq);
            const f32x2 d23 = __builtin_elementwise_fma(f32x2{acc[j][r][2], acc[j][r][3]}, f32x2{K4[r].z, K4[r].w} * cq,
                                                        f32x2{B4[r].z, B4[r].w} + bq);
            // distance kernels: sc = +distance (u = -sc), extremum = minimum; the others: sc = u, maximum
            sc[r][0] = SF::finish_abs(d01.x);
            sc[r][1] = SF::finish_abs(d01.y);
            sc[r][2] = SF::finish_abs(d23.x);
            sc[r][3] = SF::finish_abs(d23.y);
        }
        NW_PSTAMP(2);
        constexpr float WORST = SF::DIST ? INFINITY : -INFINITY;
        if (s0 + BS > N) {  // only the last support tile has rows past the bank
#pragma unroll
            for (int r = 0; r < RS; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (s0 + 16 * r + 4 * g + e >= N) sc[r][e] = WORST;
        }
        // tile-local extremum: independent chains, then the wave's four lane groups
        auto best = [](float a, float b) { return SF::DIST ? fminf(a, b) : fmaxf(a, b); };
        float mx[4] = {WORST, WORST, WORST, WORST};
#pragma unroll
        for (int r = 0; r < RS; ++r) mx[r & 3] = best(mx[r & 3], best(best(sc[r][0], sc[r][1]), best(sc[r][2], sc[r][3])));
        float ext = best(best(mx[0], mx[1]), best(mx[2], mx[3]));
        ext = SF::DIST ? group4_min(ext) : group4_max(ext);
        const float mloc = SF::DIST ? -ext : ext;  // the tile maximum of u
#pragma unroll
        for (int r = 0; r < RS; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e)  // 2^(u - max); 2^-inf = 0 for padded rows
                sc[r][e] = __builtin_amdgcn_exp2f(SF::DIST ? ext - sc[r][e] : sc[r][e] - ext);
        NW_PSTAMP(3);
        // ---- run sums.  A run is a RANGE of tile rows, so with the (wave-uniform) first rows of runs 1
        // and 2 in hand the membership of row t is a clamped difference: [t < b] = clamp(b - t, 0, 1).
        // Up to three runs (a class-sorted bank: one or two per tile) that is 2-7 VALU ops per element
        // and no dependent matrix-core chain; more runs go through the indicator MFMAs.
        float dloc;
        auto red4 = [](float x) { return group4_sum(x); };
        if (nrun <= 3) {
            float S0[2] = {0.f, 0.f}, S1[2] = {0.f, 0.f}, S2[2] = {0.f, 0.f};
            if (nrun == 1) {
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    S0[r & 1] += (sc[r][0] + sc[r][1]) + (sc[r][2] + sc[r][3]);
                }
            } else {
                // A run is a range of rows and the boundaries b1 <= b2 are wave-uniform, so a 16-row block lies
                // inside ONE run unless a boundary cuts it: whole blocks add their lanes' four-row sums with a
                // scalar 0 / 1 weight (3 adds + 3 fmas per block); only the one or two blocks a boundary cuts take
                // the per-element clamped-difference weights (7 ops per element) -- 384 -> ~80 ops per query block.
                const int b1 = bnd.x, b2 = (nrun == 3) ? bnd.y : BS;
                const float L1 = (float)(b1 - 4 * g);
                const float M2 = 1.f - (float)(b2 - 4 * g);
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    const int lo = 16 * r, hi = 16 * r + 16;
                    const bool in0 = hi <= b1, in2 = lo >= b2, in1 = lo >= b1 && hi <= b2;
                    const float quad = (sc[r][0] + sc[r][1]) + (sc[r][2] + sc[r][3]);
                    S0[r & 1] = __builtin_fmaf(in0 ? 1.f : 0.f, quad, S0[r & 1]);
                    S1[r & 1] = __builtin_fmaf(in1 ? 1.f : 0.f, quad, S1[r & 1]);
                    S2[r & 1] = __builtin_fmaf(in2 ? 1.f : 0.f, quad, S2[r & 1]);
                    if (!(in0 || in1 || in2)) {
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
                }
            }
            const float s0v = red4(S0[0] + S0[1]);
            float s1v = 0.f, s2v = 0.f;
            if (nrun >= 2) s1v = red4(S1[0] + S1[1]);
            if (nrun == 3) s2v = red4(S2[0] + S2[1]);
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
            dloc = red4((dl[0] + dl[1]) + (dl[2] + dl[3]));
            // run sums on the matrix cores: indicator (A operand) x E (already in B-operand layout)
            for (int run_base = 0; run_base < nrun; run_base += 16) {
                f32x4 Pm = {0.f, 0.f, 0.f, 0.f};
                const int want = run_base + i;
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    const int4 rid = *reinterpret_cast<const int4*>(runid + 16 * r + 4 * g);
                    Pm = __builtin_amdgcn_mfma_f32_16x16x4f32(rid.x == want ? 1.f : 0.f, sc[r][0], Pm, 0, 0, 0);
                    Pm = __builtin_amdgcn_mfma_f32_16x16x4f32(rid.y == want ? 1.f : 0.f, sc[r][1], Pm, 0, 0, 0);
                    Pm = __builtin_amdgcn_mfma_f32_16x16x4f32(rid.z == want ? 1.f : 0.f, sc[r][2], Pm, 0, 0, 0);
                    Pm = __builtin_amdgcn_mfma_f32_16x16x4f32(rid.w == want ? 1.f : 0.f, sc[r][3], Pm, 0, 0, 0);
                }
                if (b < B) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int run = run_base + 4 * g + jj;
                        if (run < nrun) ws_num[((size_t)st * BS + run) * B + b] = Pm[jj];
                    }
                }
            }
        }
        NW_PSTAMP(4);
        if (g == 0 && b < B) {
            ws_m[(size_t)st * B + b] = mloc;
            ws_den[(size_t)st * B + b] = dloc;
        }
        NW_PSTAMP(5);
    }
}

// TWO = false: one workgroup per CU, 4-buffer ring, double-buffered fragments (<= 256 VGPRs).
// TWO = true : two workgroups per CU (3-buffer ring: 80 KB of LDS each; single-buffered fragments:
//              <= 128 VGPRs), so one's epilogue and LDS-read latency run under the other's MFMAs.
// QB = 2     : 128-query tiles (one workgroup per CU, single-buffered fragments): per flop 1/3 fewer
//              bytes through the LDS fill and 44 % fewer LDS read bytes than the 64-query tile -- on
//              this power-limited loop (header comment) energy per flop is what sets the pace.
#if defined(NW_ABL_QREG) && !defined(NW_ABL_NOQ)
#define NW_ABL_NOQ
#endif
template <int RS, int KIND, bool TWO, int QB>
__global__ __launch_bounds__(TILE_THREADS, TWO ? 4 : 2) void nw_fused_f16p_kernel(
    const float* __restrict__ q, const float* __restrict__ s, const float* __restrict__ s_norm2,
    const float* __restrict__ s_scale, const float* __restrict__ q_norm2, const float* __restrict__ q_scale,
    const float* __restrict__ logit_scale, const int* __restrict__ ws_runid, const int* __restrict__ ws_nrun, const int* __restrict__ ws_bnd,
    float* __restrict__ ws_m, float* __restrict__ ws_den, float* __restrict__ ws_num, int B, int N, int d,
    int n_stiles, int n_qtiles, int qg) {
    using P = PCfg<RS, QB>;
    constexpr int BS = P::BS, BQP = P::BQP, TILE_F4 = P::TILE_F4, NI = P::NI, NI_LO = P::NI_LO, NT = P::NT;
    constexpr bool SINGLE = TWO || QB > 1;  // single-buffered fragments
    static_assert(!(TWO && QB > 1), "two 128-query workgroups do not fit the LDS of a CU");
    static_assert(BQP % (8 * NLOAD) == 0, "query and support pieces must not share a loader round");
    constexpr int NB = TWO ? 3 : 4;  // ring depth
#ifdef NW_ABL_AHEAD   // timing experiment (tools/bench_fused.hip): fewer stages in flight
    constexpr int AHEAD = NW_ABL_AHEAD;
#else
    constexpr int AHEAD = NB - 1;    // stages in flight per loader wave
#endif
    static_assert(RS > 5, "the persistent kernel is built for the tall tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hdr0 = reinterpret_cast<float*>(smem);  // NHB header buffers of HDR_F floats, by tile index mod NHB
    float4* stage = reinterpret_cast<float4*>(smem + P::HDR_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = d / BK;
    // ---- tile order.  Workgroup b runs on XCD b % 8 (round-robin dispatch, one workgroup per CU), and
    // every XCD has its own 4 MiB L2, so each XCD walks its OWN list of tiles in an order that keeps
    // its working set in that L2: XCD x owns the support tiles st = x (mod 8); its list is cut into
    // groups of `qg` query tiles (kept resident: qg * BQP * 2 KB at d = 512), and inside a group runs support-tile
    // major, so the n_cu workgroups of the XCD are on ~n_cu/qg support tiles x qg query tiles at any
    // time.
    // The support tiles beyond the last full round of 8 (n_stiles % 8 of them) are dealt by QUERY tile
    // (qt = x mod 8) instead, so every XCD gets the same number of tiles to within n_stiles % 8: with 49
    // support tiles (a shard of the K3 bank at 8 ranks) one XCD would otherwise walk 7 and seven XCDs 6.
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3, n_cu = gridDim.x >> 3;
    const int ns_x = n_stiles >> 3;              // full rounds: support tiles st = stl * 8 + x
    const int n_full = ns_x * n_qtiles;
    const int rem = n_stiles & 7;                // leftover support tiles 8 * ns_x .. n_stiles - 1
    const int nq_x = (n_qtiles - xcd + 7) >> 3;  // query tiles of this XCD in the leftover part
    const int n_local = n_full + rem * nq_x;     // tiles of this XCD
    const int grp_tiles = qg * ns_x;
    auto decode = [&](int L, int& qt, int& st) {
        if (L >= n_full) {                       // leftover part, support-tile major
            const int r = L - n_full, j = r / nq_x;
            st = 8 * ns_x + j;
            qt = xcd + 8 * (r - j * nq_x);
            return;
        }
        const int gi = L / grp_tiles, r = L - gi * grp_tiles;
        const int g = min(qg, n_qtiles - gi * qg);
        const int stl = r / g;
        qt = gi * qg + (r - stl * g);
        st = stl * 8 + xcd;
    };

    if (wave >= NCONS) {
        // ================================ LOADER ================================
        const int lw = wave - NCONS;
        const bool long_wave = (NI == NI_LO) || (lw < NT % NLOAD);
        unsigned voff[NI];
        int iT = cu, ikt = 0, irot = 0, ipar = 0;  // issue cursor: (tile of this XCD's list, stage), header buffer
        int iq0 = 0, is0 = 0, ist = 0;
        int gs = 0;                                // stages issued so far (ring position)
        auto set_tile = [&](int T) {
            int qt, st;
            decode(T, qt, st);
            iq0 = qt * BQP;
            is0 = st * BS;
            ist = st;
            irot = st % nk;
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                const int R = 8 * (lw + NLOAD * m) + (lane >> 3);
                const int lslot = (lane & 7) ^ ((R >> 1) & 7);
                // relative to the tile's first rows (64-bit bases in issue_next): no 4 GB limit on the bank
                const int rel = (8 * NLOAD * m < BQP) ? min(iq0 + R, B - 1) - iq0 : min(is0 + R - BQP, N - 1) - is0;
                voff[m] = ((unsigned)rel * (unsigned)d + lslot * 4) * 4u;
            }
        };
        auto dma4 = [&](const void* src, float* dst) {  // one dword per lane -> dst[lane]
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
        };
        // header pieces of the tile under the issue cursor: HPW per loader wave (piece ids past the
        // last one repeat the last piece: same bytes to the same place)
        auto issue_header = [&]() {
            float* h = hdr0 + ipar * P::HDR_F;
#pragma unroll
            for (int k = 0; k < P::HPW; ++k) {
                const int pc = min(lw + NLOAD * k, P::NP - 1);
                if (pc < 3 * P::N64) {
                    const int arr = pc / P::N64, c = pc - arr * P::N64;
                    const int row = is0 + 64 * c + lane;
                    float* dst = h + arr * P::NH + 64 * c;
                    if (arr == 0) dma4(s_norm2 + min(row, N - 1), dst);
                    else if (arr == 1) dma4(s_scale + min(row, N - 1), dst);
                    else dma4(ws_runid + (size_t)ist * BS + 64 * c + lane, dst);  // padded by 64 entries
                } else {
                    const int qp = pc - 3 * P::N64, arr = qp / QB, c = qp - arr * QB;  // qn2 pieces, then qsc pieces
                    const int row = min(iq0 + 64 * c + lane, B - 1);
                    dma4((arr == 0 ? q_norm2 : q_scale) + row, h + 3 * P::NH + arr * BQP + 64 * c);
                }
            }
        };
        bool young_hdr = false;  // does the youngest issued stage carry header pieces?
        auto issue_next = [&]() {  // returns false once every stage of every tile has been issued
            if (iT >= n_local) return false;
            int kc = ikt + irot;
            if (kc >= nk) kc -= nk;
            float4* buf = stage + ((unsigned)gs % NB) * TILE_F4;
            const char* qb = reinterpret_cast<const char*>(q + (size_t)iq0 * d) + (size_t)kc * BK * 4;
            const char* sb = reinterpret_cast<const char*>(s + (size_t)is0 * d) + (size_t)kc * BK * 4;
#ifndef NW_ABL_NODMA
#pragma unroll
#endif
            for (int m = 0; m < NI; ++m) {
                if (NI != NI_LO && m == NI - 1 && lw + NLOAD * m >= NT) break;
#if defined(NW_ABL_NOQ)   // timing experiment: the query rows are not filled (results wrong, half of the LDS-DMA bytes gone)
                if (8 * NLOAD * m < BQP) continue;
#endif
#ifdef NW_ABL_NOS   // ... or the support rows are not
                if (8 * NLOAD * m >= BQP) continue;
#endif
                const char* g = ((8 * NLOAD * m < BQP) ? qb : sb) + voff[m];
#ifndef NW_ABL_NODMA
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(buf + 64 * (lw + NLOAD * m)),
                                                 16, 0, 0);
#else
                (void)g; (void)buf;
#endif
            }
            young_hdr = (ikt == 0);
            if (young_hdr) issue_header();  // after the stage's own pieces: they are waited for last
            ++gs;
            if (++ikt == nk) {
                ikt = 0;
                ipar = (ipar + 1 == P::NHB) ? 0 : ipar + 1;
                iT += n_cu;
                if (iT < n_local) set_tile(iT);
            }
            return true;
        };
        auto wait_landed = [&](bool issued) {  // everything but the youngest stage of this wave has landed
#ifdef NW_ABL_AHEAD   // (with fewer stages in flight the youngest must have landed too: the consumers read one stage ahead)
            if (AHEAD < NB - 1) { wait_vmcnt<0>(); return; }
#endif
#if defined(NW_ABL_NOQ) || defined(NW_ABL_NOS)
            {
                constexpr int NQP = BQP / (8 * NLOAD);
#ifdef NW_ABL_NOQ
                constexpr int NP_ = NI - NQP;
#else
                constexpr int NP_ = NQP;
#endif
                static_assert(NI == NI_LO, "ablation builds: even split of the pieces");
                if (!issued) wait_vmcnt<0>();
                else if (young_hdr) wait_vmcnt<NP_ + P::HPW>();
                else wait_vmcnt<NP_>();
                return;
            }
#endif
            if (!issued) wait_vmcnt<0>();
            else if (long_wave) { if (young_hdr) wait_vmcnt<NI + P::HPW>(); else wait_vmcnt<NI>(); }
            else { if (young_hdr) wait_vmcnt<NI_LO + P::HPW>(); else wait_vmcnt<NI_LO>(); }
        };
        if (iT < n_local) set_tile(iT);
        bool more = true;
#pragma unroll
        for (int k0 = 0; k0 < AHEAD; ++k0) more = issue_next();
        wait_landed(more);
        tile_barrier();  // P: all but the youngest issued stage (and the first tile's header) have landed
        for (int T = cu; T < n_local; T += n_cu) {
            for (int kt = 0; kt < nk; ++kt) {
                more = issue_next();
                wait_landed(more);
                tile_barrier();
            }
        }
    } else {
        // ================================ CONSUMER ================================
        const int i = lane & 15, g = lane >> 4;
        struct Frag {
            float4 bh[QB], bl[QB];
            float4 ah[RS], al[RS];
        };
        const int rsw = (i >> 1) & 7;
        auto load_frags = [&](Frag& f, int buf) {
            const float4* Qs = stage + ((unsigned)buf % NB) * TILE_F4;
            const float4* Ss = Qs + BQP * ROW_F4;
            const int sh = g ^ rsw, sl = (4 + g) ^ rsw;
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const int qrow = 16 * (QB * wave + j) + i;  // (qrow >> 1) & 7 == rsw: 16-row blocks keep the swizzle
                f.bh[j] = Qs[qrow * ROW_F4 + sh];
                f.bl[j] = Qs[qrow * ROW_F4 + sl];
            }
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                f.ah[r] = Ss[(16 * r + i) * ROW_F4 + sh];
                f.al[r] = Ss[(16 * r + i) * ROW_F4 + sl];
            }
        };
        auto mm = [](const float4& a, const float4& b, f32x4 c) {
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
        };
        f32x4 acc[QB][RS];
        auto mfma_stage = [&](const Frag& f) {
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int r = 0; r < RS; ++r) acc[j][r] = mm(f.al[r], f.bh[j], acc[j][r]);
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int r = 0; r < RS; ++r) acc[j][r] = mm(f.ah[r], f.bl[j], acc[j][r]);
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int r = 0; r < RS; ++r) acc[j][r] = mm(f.ah[r], f.bh[j], acc[j][r]);
        };
        auto interleave = [&]() {  // QB == 1 double-buffered loop: MFMA / ds_read alternation
#pragma unroll
            for (int x = 0; x < 2 * (RS + 1); ++x) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * RS - 2 * (RS + 1), 0);
        };

        tile_barrier();  // P
        int gi = 0;      // ring position of the current tile's first stage
        int par = 0;     // header buffer of the current tile
#ifdef NW_DIAG_FUSED
        unsigned long long diag_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
        const unsigned long long first_ = last_, first_rt_ = __builtin_amdgcn_s_memrealtime();
#endif
        for (int T = cu; T < n_local; T += n_cu) {
            int qt, st;
            decode(T, qt, st);
            const int q0 = qt * BQP, s0 = st * BS;
            const int nrun = ws_nrun[st];  // wave-uniform: scalar loads, used after the main loop
            const int2 bnd = *reinterpret_cast<const int2*>(ws_bnd + 2 * (size_t)st);  // first rows of runs 1 and 2
            // (an L2 prefetch of the next tile's support rows was measured and dropped: 342 us with, 329 us
            //  without at B=2048 N=50000 d=512 -- the three-stage LDS pipeline already covers the first touch)

#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int r = 0; r < RS; ++r) acc[j][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (QB > 1) {
                // Hand-ordered stage.  The operands of a stage's first MFMA group (al, bh: F1) are double-
                // buffered and read one stage ahead, one ds_read behind each of the previous stage's first ten
                // MFMAs; those of the other two groups (ah, bl: F2) are single-buffered and re-read for the next
                // stage as the last group frees them (bl at its start, ah[r] behind the pair that used it), so
                // every LDS read has at least the 16 MFMAs of a group between its issue and its first use and
                // nothing is waited for at the barrier: the reads still in flight there belong to the NEXT
                // stage's buffer, and every read of the buffer the barrier releases has been consumed by an MFMA
                // issued in front of it.  (hipcc, left to itself, sank the fragment reads to the end of the
                // stage, in front of the barrier's lgkmcnt(0), and waited for two fresh reads at the top of the
                // stage: two exposed LDS latencies per stage, MFMA pipe idle -- 1211 ticks per stage for 768
                // cycles of MFMA.)  sched_barrier(0) after every step pins the order.
                struct F1 { float4 bh[QB]; float4 al[RS]; };
                struct F2 { float4 bl[QB]; float4 ah[RS]; };
                const int sh = g ^ rsw, sl = (4 + g) ^ rsw;
                const int qoff = (16 * QB * wave + i) * ROW_F4, soff = (BQP + i) * ROW_F4;
                auto stage_base = [&](int buf) { return stage + ((unsigned)buf % NB) * TILE_F4; };
                auto rd_bh = [&](const float4* S, int j) { return S[qoff + 16 * j * ROW_F4 + sh]; };
                auto rd_bl = [&](const float4* S, int j) { return S[qoff + 16 * j * ROW_F4 + sl]; };
                auto rd_ah = [&](const float4* S, int r) { return S[soff + 16 * r * ROW_F4 + sh]; };
                auto rd_al = [&](const float4* S, int r) { return S[soff + 16 * r * ROW_F4 + sl]; };
                auto pin = []() { __builtin_amdgcn_sched_barrier(0); };
                // one stage: ac = this stage's F1, an = the next stage's (filled here), b = this stage's F2 on
                // entry, the next stage's on exit
#ifdef NW_ABL_QREG   // timing experiment: the wave's own query rows, global -> VGPR, two stages ahead (values unused)
                const char* qtile_ = reinterpret_cast<const char*>(q + (size_t)q0 * d);
                unsigned qlane_[QB];
#pragma unroll
                for (int j = 0; j < QB; ++j)
                    qlane_[j] = ((unsigned)(min(q0 + 16 * (QB * wave + j) + i, B - 1) - q0) * (unsigned)d + 4 * g) * 4u;
                const int krot_ = st % nk;
                auto qload_ = [&](float4 (&X)[2 * QB], int kt_) {
                    int kc = min(kt_, nk - 1) + krot_;
                    if (kc >= nk) kc -= nk;
                    const char* b_ = qtile_ + (size_t)kc * (BK * 4);
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        X[2 * j] = *reinterpret_cast<const float4*>(b_ + qlane_[j]);
                        X[2 * j + 1] = *reinterpret_cast<const float4*>(b_ + qlane_[j] + 64);
                    }
                };
                float4 qx0_[2 * QB], qx1_[2 * QB];
                qload_(qx0_, 0);
                qload_(qx1_, 1);
                int kt_abs_ = 0;
#define NW_QREG_STEP(X)                                                                                   \
    do {                                                                                                  \
        _Pragma("unroll") for (int e_ = 0; e_ < 2 * QB; ++e_) asm volatile("" ::"v"(__builtin_bit_cast(f32x4, X[e_])));             \
        qload_(X, kt_abs_ + 2);                                                                           \
        ++kt_abs_;                                                                                        \
        pin();                                                                                            \
    } while (0)
#else
#define NW_QREG_STEP(X)
#endif
                auto run_stage = [&](const F1& ac, F1& an, F2& b, int buf_next, auto has_next) {
#ifdef NW_ABL_NORD   // timing experiment: no fragment reads in the loop (the first stage's fragments are reused)
                    constexpr bool NXT = false;
#else
                    constexpr bool NXT = decltype(has_next)::value;
#endif
                    const float4* Sn = stage_base(buf_next);
                    int n = 0;
#pragma unroll
                    for (int r = 0; r < RS; ++r)
#pragma unroll
                        for (int jx = 0; jx < QB; ++jx) {       // group 1: al x bh
                            const int j = (r & 1) ? QB - 1 - jx : jx;
                            acc[j][r] = mm(ac.al[r], ac.bh[j], acc[j][r]);
                            if (NXT && n < QB + RS) {
                                if (n < QB) an.bh[n] = rd_bh(Sn, n);
                                else an.al[n - QB] = rd_al(Sn, n - QB);
                            }
                            ++n;
                            pin();
                        }
#pragma unroll
                    for (int r = 0; r < RS; ++r)
#pragma unroll
                        for (int jx = 0; jx < QB; ++jx) {       // group 2: ah x bl
                            const int j = (r & 1) ? QB - 1 - jx : jx;
                            acc[j][r] = mm(b.ah[r], b.bl[j], acc[j][r]);
                            pin();
                        }
                    if (NXT) {
#pragma unroll
                        for (int j = 0; j < QB; ++j) b.bl[j] = rd_bl(Sn, j);
                        pin();
                    }
#pragma unroll
                    for (int r = RS - 1; r >= 0; --r) {         // group 3: ah x bh, ah[r] re-read behind its pair
#pragma unroll
                        for (int jx = 0; jx < QB; ++jx) {
                            const int j = (r & 1) ? jx : QB - 1 - jx;
                            acc[j][r] = mm(b.ah[r], ac.bh[j], acc[j][r]);
                        }
                        pin();
                        if (NXT) {
                            b.ah[r] = rd_ah(Sn, r);
                            pin();
                        }
                    }
                };
                using Yes = std::integral_constant<bool, true>;
                using No = std::integral_constant<bool, false>;
                F1 a0, a1;
                F2 b0;
                {
                    const float4* S0 = stage_base(gi);
#pragma unroll
                    for (int j = 0; j < QB; ++j) { a0.bh[j] = rd_bh(S0, j); b0.bl[j] = rd_bl(S0, j); }
#pragma unroll
                    for (int r = 0; r < RS; ++r) { a0.al[r] = rd_al(S0, r); b0.ah[r] = rd_ah(S0, r); }
#ifdef NW_ABL_NORD
                    a1 = a0;
#endif
                    pin();
                }
                int kt = 0;
                for (; kt + 2 < nk; kt += 2) {
                    NW_QREG_STEP(qx0_);
                    run_stage(a0, a1, b0, gi + kt + 1, Yes{});
                    tile_barrier_nowait();
                    NW_QREG_STEP(qx1_);
                    run_stage(a1, a0, b0, gi + kt + 2, Yes{});
                    tile_barrier_nowait();
                }
                if (kt + 2 == nk) {
                    NW_QREG_STEP(qx0_);
                    run_stage(a0, a1, b0, gi + kt + 1, Yes{});
                    tile_barrier_nowait();
                    NW_QREG_STEP(qx1_);
                    run_stage(a1, a0, b0, 0, No{});
                    tile_barrier_nowait();
                } else {
                    NW_QREG_STEP(qx0_);
                    run_stage(a0, a1, b0, 0, No{});
                    tile_barrier_nowait();
                }
#undef NW_QREG_STEP
            } else if constexpr (SINGLE) {
                Frag f0;
                for (int kt = 0; kt < nk; ++kt) {
                    load_frags(f0, gi + kt);
                    mfma_stage(f0);
                    tile_barrier();
                }
            } else {
                Frag f0, f1;
                load_frags(f0, gi);
                int kt = 0;
                for (; kt + 2 < nk; kt += 2) {
                    load_frags(f1, gi + kt + 1);
                    mfma_stage(f0);
                    interleave();
                    tile_barrier();
                    load_frags(f0, gi + kt + 2);
                    mfma_stage(f1);
                    interleave();
                    tile_barrier();
                }
                for (; kt < nk; ++kt) {
                    if (kt + 1 < nk) load_frags(f1, gi + kt + 1);
                    mfma_stage(f0);
                    tile_barrier();
                    f0 = f1;
                }
            }
            gi += nk;
            NW_PSTAMP(0);
#ifndef NW_ABL_NOEPI
            epilogue_p<RS, KIND, QB>(acc, hdr0 + par * P::HDR_F, nrun, bnd, logit_scale, ws_m, ws_den, ws_num, B, N, q0, s0, st,
                                 wave, lane
#ifdef NW_DIAG_FUSED
                                 , diag_, last_
#endif
                                 );
#else
            {  // ablation build: keep every accumulator chain alive
                f32x4 sum_ = {0.f, 0.f, 0.f, 0.f};
                for (int j = 0; j < QB; ++j)
                    for (int r = 0; r < RS; ++r) sum_ += acc[j][r];
                if (sum_[0] + sum_[1] + sum_[2] + sum_[3] == 12345.678f) ws_m[tid] = sum_[0] + nrun + bnd.x;
            }
#endif
            par = (par + 1 == P::NHB) ? 0 : par + 1;
            NW_PSTAMP(6);
        }
#ifdef NW_DIAG_FUSED
        if (tid == 0 && blockIdx.x < 1024) {
            for (int k = 0; k < 7; ++k) nw_diag_p[8 * blockIdx.x + k] = diag_[k];
            nw_diag_p[8 * blockIdx.x + 7] = last_ - first_;
            nw_diag_rt[2 * blockIdx.x] = first_rt_;
            nw_diag_rt[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        }
#endif
    }
}

}  // namespace
}  // namespace nw
