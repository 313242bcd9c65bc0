// fused_f16p.h -- persistent edition of the split-fp16 fused forward (gfx950 / MI355X only).
//
// One 512-thread workgroup per CU walks its share of the (query tile, support tile) grid
// (tile T = blockIdx.x, + gridDim.x, ...), instead of one workgroup per tile.  Measured on the
// one-workgroup-per-tile kernel (tools/bench_fused.hip, B=2048 N=50000): 21.9 k cycles of stamped
// work per tile but 27 k cycles per tile of wall time: every new workgroup pays its dispatch (LDS
// allocation, wave launch), the latency of its label/norm loads and the DMA prologue before the first
// MFMA.  Here
//   * the loader waves keep ONE stage pipeline running across tile boundaries: while the consumers
//     run a tile's epilogue the first three stages of the next tile are already landing;
//   * the consumers issue the loads for a tile's header (labels, norms, scales) when the tile starts
//     and only touch them after its main loop, so that latency hides behind the MFMAs;
//   * barriers: one per stage plus one per tile (header visible), executed by both roles.
// Everything else (tile shape, LDS image, MFMA stream, epilogue, workspace layout) is that of
// nw_fused_kernel<RS, KIND, false, MODE_F16>.
#pragma once
#include "fused_impl.h"

namespace nw {
namespace {

#ifdef NW_DIAG_FUSED  // diagnostic build only (tools/bench_fused.hip): per-workgroup phase totals
__device__ unsigned long long nw_diag_p[8 * 1024];
#define NW_PSTAMP(k)                                              \
    do {                                                          \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        diag_[k] += now_ - last_;                                 \
        last_ = now_;                                             \
    } while (0)
#else
#define NW_PSTAMP(k)
#endif

template <int RS, int KIND>
__global__ __launch_bounds__(TILE_THREADS, 2) void nw_fused_f16p_kernel(
    const float* __restrict__ q, const float* __restrict__ s, const int64_t* __restrict__ sy,
    const float* __restrict__ s_norm2, const float* __restrict__ s_scale, const float* __restrict__ q_norm2,
    const float* __restrict__ q_scale, const float* __restrict__ logit_scale, float* __restrict__ ws_m,
    float* __restrict__ ws_den, int* __restrict__ ws_nrun, int* __restrict__ ws_lab,
    float* __restrict__ ws_num, int B, int N, int d, int C, int n_stiles, int n_qtiles, int qg) {
    using Cfg = DmaCfg<RS>;
    constexpr int BS = Cfg::BS, TILE_F4 = Cfg::TILE_F4, NI = Cfg::NI, NI_LO = Cfg::NI_LO, NT = Cfg::NT;
    constexpr int AHEAD = Cfg::NBUF - 1;
    constexpr bool NEED_NORM = (KIND != NW_SCORE_DOT);
    static_assert(RS > 5, "the persistent kernel uses the double-buffered fragment loop");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qn2 = reinterpret_cast<float*>(smem);
    float* qsc_s = qn2 + 64;
    float* sn2 = qsc_s + 64;
    float* ssc = sn2 + RUN_CAP;
    int* runid = reinterpret_cast<int*>(ssc + RUN_CAP);
    int* runlab = runid + RUN_CAP;
    int* nrun_s = runlab + RUN_CAP;
    float4* stage = reinterpret_cast<float4*>(smem + FUSED_HDR);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = d / BK;
    // ---- tile order.  Workgroup b runs on XCD b % 8 (round-robin dispatch, one workgroup per CU), and
    // every XCD has its own 4 MiB L2, so each XCD walks its OWN list of tiles in an order that keeps
    // its working set in that L2: XCD x owns the support tiles st = x (mod 8); its list is cut into
    // groups of `qg` query tiles (kept resident: qg * 128 KB), and inside a group runs support-tile
    // major, so the n_cu workgroups of the XCD are on ~n_cu/qg support tiles x qg query tiles at any
    // time.  (Measured with the old order -- all 32 query tiles of ONE support tile per XCD, 4 MB of
    // queries cycling through L2 -- the LDS fill ran at 65 GB/s per CU and set the pace of the main
    // loop; from an L2-resident source the same loop streams 140 GB/s per CU: tools/bench_f16_loop.)
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3, n_cu = gridDim.x >> 3;
    const int ns_x = (n_stiles - xcd + 7) >> 3;  // support tiles of this XCD
    const int n_local = ns_x * n_qtiles;         // tiles of this XCD
    const int grp_tiles = qg * ns_x;
    int dec_g = 1, dec_c = 0;  // of the last decode: workgroups sharing the support tile, and this one's rank
    auto decode = [&](int L, int& qt, int& st) {
        const int gi = L / grp_tiles, r = L - gi * grp_tiles;
        const int g = min(qg, n_qtiles - gi * qg);
        const int stl = r / g;
        dec_g = g;
        dec_c = r - stl * g;
        qt = gi * qg + dec_c;
        st = stl * 8 + xcd;
    };
    if (wave >= NCONS) {
        // ================================ LOADER ================================
        const int lw = wave - NCONS;
        const bool long_wave = (NI == NI_LO) || (lw < NT % NLOAD);
        unsigned voff[NI];
        int iT = cu, ikt = 0, irot = 0;  // issue cursor: (tile of this XCD's list, stage)
        int gs = 0;                                           // stages issued so far (ring position)
        auto set_tile = [&](int T) {
            int qt, st;
            decode(T, qt, st);
            const int q0 = qt * BQ, s0 = st * BS;
            irot = st % nk;
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                const int R = 8 * (lw + NLOAD * m) + (lane >> 3);
                const int lslot = (lane & 7) ^ ((R >> 1) & 7);
                const int grow = (8 * NLOAD * m < BQ) ? min(q0 + R, B - 1) : min(s0 + R - BQ, N - 1);
                voff[m] = ((unsigned)grow * (unsigned)d + lslot * 4) * 4u;
            }
        };
        auto issue_next = [&]() {  // returns false once every stage of every tile has been issued
            if (iT >= n_local) return false;
            int kc = ikt + irot;
            if (kc >= nk) kc -= nk;
            float4* buf = stage + (gs & (Cfg::NBUF - 1)) * TILE_F4;
            const char* qb = reinterpret_cast<const char*>(q) + (size_t)kc * BK * 4;
            const char* sb = reinterpret_cast<const char*>(s) + (size_t)kc * BK * 4;
#ifndef NW_ABL_NODMA
#pragma unroll
#endif
            for (int m = 0; m < NI; ++m) {
                if (NI != NI_LO && m == NI - 1 && lw + NLOAD * m >= NT) break;
                const char* g = ((8 * NLOAD * m < BQ) ? qb : sb) + voff[m];
#ifndef NW_ABL_NODMA
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(buf + 64 * (lw + NLOAD * m)),
                                                 16, 0, 0);
#else
                (void)g; (void)buf;
#endif
            }
            ++gs;
            if (++ikt == nk) {
                ikt = 0;
                iT += n_cu;
                if (iT < n_local) set_tile(iT);
            }
            return true;
        };
        auto wait_landed = [&](bool issued) {  // all but the youngest stage of this wave's DMAs have landed
            if (!issued) wait_vmcnt<0>();
            else if (long_wave) wait_vmcnt<(AHEAD - 2) * NI>();
            else wait_vmcnt<(AHEAD - 2) * NI_LO>();
        };
        if (iT < n_local) set_tile(iT);
        bool more = true;
#pragma unroll
        for (int k0 = 0; k0 < AHEAD; ++k0) more = issue_next();
        wait_landed(more);
        tile_barrier();  // P: stages 0 and 1 of the first tile have landed
        for (int T = cu; T < n_local; T += n_cu) {
            for (int kt = 0; kt < nk; ++kt) {
                more = issue_next();
                wait_landed(more);
                tile_barrier();
            }
            tile_barrier();  // H: matches the consumers' "header visible" barrier
        }
    } else {
        // ================================ CONSUMER ================================
        const int i = lane & 15, g = lane >> 4;
        struct Frag {
            float4 bh, bl;
            float4 ah[RS], al[RS];
        };
        const int qrow = 16 * wave + i;
        const int rsw = (i >> 1) & 7;
        auto load_frags = [&](Frag& f, int buf) {
            const float4* Qs = stage + (buf & (Cfg::NBUF - 1)) * TILE_F4;
            const float4* Ss = Qs + BQ * ROW_F4;
            const int sh = g ^ rsw, sl = (4 + g) ^ rsw;
            f.bh = Qs[qrow * ROW_F4 + sh];
            f.bl = Qs[qrow * ROW_F4 + sl];
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                f.ah[r] = Ss[(16 * r + i) * ROW_F4 + sh];
                f.al[r] = Ss[(16 * r + i) * ROW_F4 + sl];
            }
        };
        auto mm = [](const float4& a, const float4& b, f32x4 c) {
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
        };
        f32x4 acc[RS];
        auto mfma_stage = [&](const Frag& f) {
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = mm(f.al[r], f.bh, acc[r]);
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = mm(f.ah[r], f.bl, acc[r]);
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = mm(f.ah[r], f.bh, acc[r]);
        };
        auto interleave = [&]() {
#pragma unroll
            for (int x = 0; x < 2 * (RS + 1); ++x) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * RS - 2 * (RS + 1), 0);
        };

        tile_barrier();  // P
        int gi = 0;      // ring position of the current tile's first stage
#ifdef NW_DIAG_FUSED
        unsigned long long diag_[4] = {0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
        const unsigned long long first_ = last_;
#endif
        for (int T = cu; T < n_local; T += n_cu) {
            int qt, st;
            decode(T, qt, st);
            const int q0 = qt * BQ, s0 = st * BS;
            // header loads issued now, consumed behind the main loop: two values per thread + labels
            const int t0 = tid, t1 = tid + 64 * NCONS;  // 256 consumer threads cover BS <= 192 rows... twice
            float h_sn0 = 0.f, h_ss0 = 0.f, h_qn = 0.f, h_qs = 1.f;
            int lab[3] = {-1, -1, -1};
            if (t0 < BS) {
                if (NEED_NORM) h_sn0 = s_norm2[min(s0 + t0, N - 1)];
                h_ss0 = s_scale[min(s0 + t0, N - 1)];
            }
            if (t0 < BQ) {
                if (NEED_NORM) h_qn = q_norm2[min(q0 + t0, B - 1)];
                h_qs = q_scale[min(q0 + t0, B - 1)];
            }
            (void)t1;
            if (wave == 0) load_tile_labels<BS>(sy, s0, N, C, lane, lab);
#ifndef NW_ABL_NOPREFETCH
            // L2 prefetch: the support rows of this workgroup's NEXT tile, one 128-B line per load, the
            // lines dealt round-robin to the workgroups that will share that tile.  First touches of
            // support rows otherwise pay the HBM / Infinity Cache latency inside the 3-stage LDS
            // pipeline (74 KB in flight per CU / ~1.1 us = the 65 GB/s per CU measured); touched one
            // tile ahead, the LDS-DMA stream only ever hits in L2.
            if (T + n_cu < n_local) {
                int nqt, nst;
                decode(T + n_cu, nqt, nst);
                const char* nb = reinterpret_cast<const char*>(s) + (size_t)nst * BS * d * 4;
                const int n_lines = min(BS, N - nst * BS) * nk;
                for (int x = tid * dec_g + dec_c; x < n_lines; x += 64 * NCONS * dec_g)
                    (void)*reinterpret_cast<const volatile float*>(nb + (size_t)x * 128);
            }
#endif

#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
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
            gi += nk;
            NW_PSTAMP(0);

            // header of this tile (the previous tile's epilogue is long over: every wave has passed
            // this tile's stage barriers since)
            if (t0 < BS) {
                if (NEED_NORM) sn2[t0] = h_sn0;
                ssc[t0] = h_ss0;
            }
            if (t0 < BQ) {
                if (NEED_NORM) qn2[t0] = h_qn;
                qsc_s[t0] = h_qs;
            }
            if (wave == 0) run_scan_wave<BS>(lab, lane, runid, runlab, nrun_s);
            tile_barrier();  // H
            NW_PSTAMP(1);
#ifndef NW_ABL_NOEPI
            fused_epilogue<RS, KIND, false, MODE_F16>(acc, qn2, sn2, ssc, runid, runlab, nrun_s, qsc_s, logit_scale,
                                                      nullptr, ws_m, ws_den, ws_nrun, ws_lab, ws_num, B, N, q0, s0, qt, st);
#else
            if (acc[0][0] + acc[RS - 1][3] == 12345.678f) ws_m[tid] = acc[1][1];
#endif
            NW_PSTAMP(2);
        }
#ifdef NW_DIAG_FUSED
        if (tid == 0 && blockIdx.x < 1024) {
            for (int k = 0; k < 3; ++k) nw_diag_p[8 * blockIdx.x + k] = diag_[k];
            nw_diag_p[8 * blockIdx.x + 3] = last_ - first_;
        }
#endif
    }
}

}  // namespace
}  // namespace nw
