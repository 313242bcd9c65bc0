// fused_f16p8.h -- persistent split-fp16 fused forward, EIGHT multiplying waves (gfx950 / MI355X only).
//
// Why (round 4, tools/bench_fused.hip ablations of nw_fused_f16p_kernel<8, 0, false, 2> on the K3 launch, ticks per 32-k stage
// of a consumer wave: 768 of them are MFMA issue): 1 100 as built; 1 074 with HALF of the LDS fill gone (queries or supports
// not loaded: the fill is not the pole); 1 031 with no fill at all; 973 without the fragment reads; 885 with neither -- bare
// MFMAs and one barrier per stage still cost 15 % over their issue time, the reads add 13-19 %, the fill 6-9 %, and the
// epilogue (29 % of the ticks) runs on ONE wave per SIMD, i.e. at half the vector issue rate, with the matrix pipe idle.
// The loader waves that share the SIMDs are idle nearly all the time.  So here every wave multiplies:
//   * workgroup tile 256 queries x 128 supports, eight waves of 32 queries x 128 supports each (the consumer wave of
//     fused_f16p.h, twice): per flop 25 % fewer bytes through the LDS fill, half the barriers, and the two waves of a
//     SIMD fill each other's stalls (fragment reads, DMA issue, barrier skew, the epilogue's latency chains);
//   * every wave issues its share of the stage's LDS-DMAs itself (6 pieces of 1 KB per stage, spread over the stage's
//     MFMAs; waves 4-7 at the slots between those of waves 0-3) and the tile header with the tile's first stage;
//   * 48 KB stages in a ring of THREE: during stage G the waves read buffer G+1, buffer G+2 lands, and the DMAs of stage
//     G+3 go into buffer G, whose fragment reads (issued during stage G-1) are waited for in front of barrier G-1;
//   * fragment reads are inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every LDS read it can see in a wave with
//     LDS-DMAs in flight.  One s_waitcnt lgkmcnt(0), tied to the fragment registers, ends each stage; the stage is ordered
//     (ah x bl, ah x bh, al x bh) so that the last read is issued 16 MFMAs before it.
// Epilogue, run tables, workspace layout and merge are those of fused_f16p.h (support tiles of 128 rows).
#pragma once
#include "fused_f16p.h"

namespace nw {
namespace {

struct P8 {
    static constexpr int RS = 8, QB = 2, NWV = 8;
    static constexpr int BS = 16 * RS;                     // 128 supports
    static constexpr int BQP = 16 * QB * NWV;              // 256 queries
    static constexpr int ROWS = BQP + BS;                  // rows of a stage image
    static constexpr int TILE_F4 = ROWS * ROW_F4;
    static constexpr int STAGE_BYTES = TILE_F4 * 16;       // 48 KB
    static constexpr int NB = 3;                           // ring depth
    static constexpr int NT = ROWS / 8;                    // 1 KB pieces per stage
    static constexpr int NI = NT / NWV;                    // ... per wave
    static constexpr int NQP = BQP / 8 / NWV;              // of them query pieces (the first ones)
    static constexpr int NH = BS;                          // entries per support-side header array
    static constexpr int HDR_F = 3 * NH + 2 * BQP;         // sn2 | ssc | runid | qn2[BQP] | qsc[BQP]
    static constexpr int NP = 3 * (NH / 64) + 2 * (BQP / 64);   // header pieces (256 B each)
    static constexpr int HPW = (NP + NWV - 1) / NWV;       // ... per wave
    static constexpr int NHB = 2;                          // header buffers (tile parity)
    static constexpr size_t HDR_BYTES = (size_t)NHB * HDR_F * 4;
    static constexpr size_t LDS_BYTES = HDR_BYTES + (size_t)NB * STAGE_BYTES;
    static_assert(NT % NWV == 0 && (BQP / 8) % NWV == 0, "even split of the pieces");
    static_assert(HDR_BYTES % 16 == 0 && LDS_BYTES <= 160 * 1024, "LDS of one CU");
    static_assert(NI + HPW < 64, "vmcnt is a 6-bit field");
};

#define NW_LDS_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int KIND>
__global__ __launch_bounds__(TILE_THREADS, 2) void nw_fused_f16p8_kernel(
    const float* __restrict__ q, const float* __restrict__ s, const float* __restrict__ s_norm2,
    const float* __restrict__ s_scale, const float* __restrict__ q_norm2, const float* __restrict__ q_scale,
    const float* __restrict__ logit_scale, const int* __restrict__ ws_runid, const int* __restrict__ ws_nrun,
    const int* __restrict__ ws_bnd, float* __restrict__ ws_m, float* __restrict__ ws_den, float* __restrict__ ws_num, int B,
    int N, int d, int n_stiles, int n_qtiles, int qg) {
    using P = P8;
    constexpr int RS = P::RS, QB = P::QB, BS = P::BS, BQP = P::BQP, NI = P::NI, NB = P::NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hdr0 = reinterpret_cast<float*>(smem);
    float4* stage = reinterpret_cast<float4*>(smem + P::HDR_BYTES);
    const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + P::HDR_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = wave >> 2;          // waves w and w + 4 share a SIMD: they take alternate DMA slots
    const int nk = d / BK;
    // ---- tile order: as nw_fused_f16p_kernel (XCD-local lists, groups of qg query tiles, support-tile major)
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3, n_cu = gridDim.x >> 3;
    const int ns_x = n_stiles >> 3;
    const int n_full = ns_x * n_qtiles;
    const int rem = n_stiles & 7;
    const int nq_x = (n_qtiles - xcd + 7) >> 3;
    const int n_local = n_full + rem * nq_x;
    const int grp_tiles = qg * ns_x;
    auto decode = [&](int L, int& qt, int& st) {
        if (L >= n_full) {
            const int r = L - n_full, j = r / nq_x;
            st = 8 * ns_x + j;
            qt = xcd + 8 * (r - j * nq_x);
            return;
        }
        const int gi_ = L / grp_tiles, r = L - gi_ * grp_tiles;
        const int g_ = min(qg, n_qtiles - gi_ * qg);
        const int stl = r / g_;
        qt = gi_ * qg + (r - stl * g_);
        st = stl * 8 + xcd;
    };

    // ================================ the wave's share of the LDS fill ================================
    unsigned voff[NI];
    int iT = cu, ikt = 0, irot = 0, ipar = 0;     // issue cursor: (tile of this XCD's list, stage), header buffer
    int iq0 = 0, is0 = 0, ist = 0;
    int gs = 0;                                   // ring slot of the stage under the cursor
    auto set_tile = [&](int T) {
        int qt, st;
        decode(T, qt, st);
        iq0 = qt * BQP;
        is0 = st * BS;
        ist = st;
        irot = st % nk;
#pragma unroll
        for (int m = 0; m < NI; ++m) {
            const int R = 8 * (wave + P::NWV * m) + (lane >> 3);      // row of the stage image: queries, then supports
            const int lslot = (lane & 7) ^ ((R >> 1) & 7);           // swizzle on the source side (an LDS-DMA writes linearly)
            const int rel = (m < P::NQP) ? min(iq0 + R, B - 1) - iq0 : min(is0 + R - BQP, N - 1) - is0;
            voff[m] = ((unsigned)rel * (unsigned)d + lslot * 4) * 4u;
        }
    };
    const char* cur_qb = nullptr;
    const char* cur_sb = nullptr;
    float4* cur_buf = nullptr;
    bool cur_live = false, cur_hdr = false;
    auto begin_issue = [&]() {            // bases of the stage under the cursor (scalar work)
        cur_live = iT < n_local;
        if (!cur_live) return;
        int kc = ikt + irot;
        if (kc >= nk) kc -= nk;
        cur_buf = stage + gs * P::TILE_F4;
        cur_qb = reinterpret_cast<const char*>(q + (size_t)iq0 * d) + (size_t)kc * BK * 4;
        cur_sb = reinterpret_cast<const char*>(s + (size_t)is0 * d) + (size_t)kc * BK * 4;
        cur_hdr = (ikt == 0);
    };
    auto issue_piece = [&](auto mc) {     // piece m of this wave
        constexpr int m = decltype(mc)::value;
        if (!cur_live) return;
#ifdef NW_ABL_NODMA
        return;
#endif
        const char* g = ((m < P::NQP) ? cur_qb : cur_sb) + voff[m];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(cur_buf + 64 * (wave + P::NWV * m)), 16, 0, 0);
    };
    auto dma4 = [&](const void* src, float* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
    };
    auto end_issue = [&]() {              // header pieces of a tile's first stage, then the cursor moves on
        if (!cur_live) return;
#ifdef NW_ABL_NODMA
        cur_hdr = false;
#endif
        if (cur_hdr) {
            float* h = hdr0 + ipar * P::HDR_F;
#pragma unroll
            for (int k = 0; k < P::HPW; ++k) {
                const int pc = min(wave + P::NWV * k, P::NP - 1);
                if (pc < 3 * (P::NH / 64)) {
                    const int arr = pc / (P::NH / 64), c = pc - arr * (P::NH / 64);
                    const int row = is0 + 64 * c + lane;
                    float* dst = h + arr * P::NH + 64 * c;
                    if (arr == 0) dma4(s_norm2 + min(row, N - 1), dst);
                    else if (arr == 1) dma4(s_scale + min(row, N - 1), dst);
                    else dma4(ws_runid + (size_t)ist * BS + 64 * c + lane, dst);   // padded by 64 entries
                } else {
                    const int qp = pc - 3 * (P::NH / 64), arr = qp / (BQP / 64), c = qp - arr * (BQP / 64);
                    const int row = min(iq0 + 64 * c + lane, B - 1);
                    dma4((arr == 0 ? q_norm2 : q_scale) + row, h + 3 * P::NH + arr * BQP + 64 * c);
                }
            }
        }
        gs = (gs + 1 == NB) ? 0 : gs + 1;
        if (++ikt == nk) {
            ikt = 0;
            ipar ^= 1;
            iT += n_cu;
            if (iT < n_local) set_tile(iT);
        }
    };
    // everything but the pieces issued by the last begin/end pair has landed (this wave's share)
    auto wait_older = [&]() {
        if (!cur_live) wait_vmcnt<0>();
        else if (cur_hdr) wait_vmcnt<NI + P::HPW>();
        else wait_vmcnt<NI>();
    };
    auto issue_whole = [&]() {
        begin_issue();
        issue_piece(std::integral_constant<int, 0>{});
        issue_piece(std::integral_constant<int, 1>{});
        issue_piece(std::integral_constant<int, 2>{});
        issue_piece(std::integral_constant<int, 3>{});
        issue_piece(std::integral_constant<int, 4>{});
        issue_piece(std::integral_constant<int, 5>{});
        static_assert(NI == 6, "pieces per wave");
        end_issue();
    };

    // ================================ fragments ================================
    const int i = lane & 15, g = lane >> 4;
    const int rsw = (i >> 1) & 7;
    // byte offsets of this lane inside a stage image: query row 32 * wave + i (+ 16 j), support row BQP + i (+ 16 r);
    // 16-byte slot g (high halves) / 4 + g (low halves), swizzled by the row
    const unsigned lq_h = (unsigned)((2 * 16 * wave + i) * ROW_F4 + (g ^ rsw)) * 16u;
    const unsigned lq_l = (unsigned)((2 * 16 * wave + i) * ROW_F4 + ((4 + g) ^ rsw)) * 16u;
    const unsigned ls_h = (unsigned)((BQP + i) * ROW_F4 + (g ^ rsw)) * 16u;
    const unsigned ls_l = (unsigned)((BQP + i) * ROW_F4 + ((4 + g) ^ rsw)) * 16u;
    constexpr int BLK = 16 * ROW_F4 * 16;   // bytes between 16-row blocks (2 KB): the reads' immediate offsets
    struct F1 { f32x4 bh[QB]; f32x4 al[RS]; };
    struct F2 { f32x4 bl[QB]; f32x4 ah[RS]; };
    auto mm = [](const f32x4& a, const f32x4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    };
    auto pin = []() { __builtin_amdgcn_sched_barrier(0); };
    f32x4 acc[QB][RS];

    // one stage: c = this stage's F1, n = the next stage's (filled here), f = this stage's F2 on entry, the next stage's
    // on exit; nbase = LDS byte address of the next stage's image.  MFMA order: ah x bl, ah x bh, al x bh.
    auto run_stage = [&](const F1& c, F1& n, F2& f, unsigned nbase, auto has_next, bool dma) {
#ifdef NW_ABL_NORD
        constexpr bool NXT = false;
#else
        constexpr bool NXT = decltype(has_next)::value;
#endif
        const unsigned aqh = nbase + lq_h, aql = nbase + lq_l, ash = nbase + ls_h, asl = nbase + ls_l;
        if (dma) begin_issue();
#ifdef NW_ABL_DMAFRONT   // timing experiment: the wave's six pieces at the top of the stage
        if (dma) {
            issue_piece(std::integral_constant<int, 0>{}); issue_piece(std::integral_constant<int, 1>{});
            issue_piece(std::integral_constant<int, 2>{}); issue_piece(std::integral_constant<int, 3>{});
            issue_piece(std::integral_constant<int, 4>{}); issue_piece(std::integral_constant<int, 5>{});
        }
        dma = false;
        const bool dma_end_ = true;
#else
        const bool dma_end_ = dma;
#endif
        // behind every fourth MFMA of the stage there is a DMA slot; waves 0-3 take the even ones, waves 4-7 (their SIMD
        // partners) the odd ones, so that the two waves of a SIMD do not sit in a DMA issue at the same time
        int mi = 0;
        // ---- group A: ah x bl; the next stage's F1 (double-buffered) is read behind its first MFMAs
#pragma unroll
        for (int r = 0; r < RS; ++r)
#pragma unroll
            for (int jx = 0; jx < QB; ++jx) {
                const int j = (r & 1) ? QB - 1 - jx : jx;
                acc[j][r] = mm(f.ah[r], f.bl[j], acc[j][r]);
                if constexpr (NXT) {
                    const int nrd = r * QB + jx;
                    if (nrd < QB) { if (nrd == 0) NW_LDS_RD128(n.bh[0], aqh, 0); else NW_LDS_RD128(n.bh[1], aqh, BLK); }
                    else if (nrd < QB + RS) {
                        switch (nrd - QB) {
                            case 0: NW_LDS_RD128(n.al[0], asl, 0 * BLK); break;
                            case 1: NW_LDS_RD128(n.al[1], asl, 1 * BLK); break;
                            case 2: NW_LDS_RD128(n.al[2], asl, 2 * BLK); break;
                            case 3: NW_LDS_RD128(n.al[3], asl, 3 * BLK); break;
                            case 4: NW_LDS_RD128(n.al[4], asl, 4 * BLK); break;
                            case 5: NW_LDS_RD128(n.al[5], asl, 5 * BLK); break;
                            case 6: NW_LDS_RD128(n.al[6], asl, 6 * BLK); break;
                            default: NW_LDS_RD128(n.al[7], asl, 7 * BLK); break;
                        }
                    }
                }
                if ((mi & 3) == 3) {
                    const int sl = mi >> 2;
                    if (dma && (sl & 1) == hi) {
                        switch (sl >> 1) {
                            case 0: issue_piece(std::integral_constant<int, 0>{}); break;
                            default: issue_piece(std::integral_constant<int, 1>{}); break;
                        }
                    }
                }
                ++mi;
                pin();
            }
        if constexpr (NXT) {   // bl is free: the next stage's
            NW_LDS_RD128(f.bl[0], aql, 0);
            NW_LDS_RD128(f.bl[1], aql, BLK);
            pin();
        }
        // ---- group B: ah x bh, ah[r] re-read behind the pair that used it last
#pragma unroll
        for (int r = RS - 1; r >= 0; --r) {
#pragma unroll
            for (int jx = 0; jx < QB; ++jx) {
                const int j = (r & 1) ? jx : QB - 1 - jx;
                acc[j][r] = mm(f.ah[r], c.bh[j], acc[j][r]);
                if ((mi & 3) == 3) {
                    const int sl = mi >> 2;
                    if (dma && (sl & 1) == hi) {
                        switch (sl >> 1) {
                            case 2: issue_piece(std::integral_constant<int, 2>{}); break;
                            default: issue_piece(std::integral_constant<int, 3>{}); break;
                        }
                    }
                }
                ++mi;
            }
            pin();
            if constexpr (NXT) {
                switch (r) {
                    case 0: NW_LDS_RD128(f.ah[0], ash, 0 * BLK); break;
                    case 1: NW_LDS_RD128(f.ah[1], ash, 1 * BLK); break;
                    case 2: NW_LDS_RD128(f.ah[2], ash, 2 * BLK); break;
                    case 3: NW_LDS_RD128(f.ah[3], ash, 3 * BLK); break;
                    case 4: NW_LDS_RD128(f.ah[4], ash, 4 * BLK); break;
                    case 5: NW_LDS_RD128(f.ah[5], ash, 5 * BLK); break;
                    case 6: NW_LDS_RD128(f.ah[6], ash, 6 * BLK); break;
                    default: NW_LDS_RD128(f.ah[7], ash, 7 * BLK); break;
                }
                pin();
            }
        }
        // ---- group C: al x bh; no reads: the last one above is 16 MFMAs old at the end of the stage
#pragma unroll
        for (int r = 0; r < RS; ++r)
#pragma unroll
            for (int jx = 0; jx < QB; ++jx) {
                const int j = (r & 1) ? QB - 1 - jx : jx;
                acc[j][r] = mm(c.al[r], c.bh[j], acc[j][r]);
                if ((mi & 3) == 3) {
                    const int sl = mi >> 2;
                    if (dma && (sl & 1) == hi) {
                        switch (sl >> 1) {
                            case 4: issue_piece(std::integral_constant<int, 4>{}); break;
                            default: issue_piece(std::integral_constant<int, 5>{}); break;
                        }
                    }
                }
                ++mi;
                pin();
            }
        if (dma_end_) end_issue();
        if constexpr (NXT) {
            // the fragment registers are written behind the compiler's back: everything that reads them later depends on this
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(n.bh[0]), "+v"(n.bh[1]), "+v"(n.al[0]), "+v"(n.al[1]), "+v"(n.al[2]), "+v"(n.al[3]), "+v"(n.al[4]),
                           "+v"(n.al[5]), "+v"(n.al[6]), "+v"(n.al[7]), "+v"(f.bl[0]), "+v"(f.bl[1]), "+v"(f.ah[0]), "+v"(f.ah[1]),
                           "+v"(f.ah[2]), "+v"(f.ah[3]), "+v"(f.ah[4]), "+v"(f.ah[5]), "+v"(f.ah[6]), "+v"(f.ah[7]));
        }
        if (dma_end_) wait_older(); // the stage issued one iteration ago has landed
        pin();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        pin();
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    // ---- prologue: three stages in flight, the first two landed
    if (iT < n_local) set_tile(iT);
    issue_whole();
    issue_whole();
    issue_whole();
    wait_older();
    pin();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    pin();

    int gi = 0;      // ring slot of the current tile's first stage
    int par = 0;     // header buffer of the current tile
#ifdef NW_DIAG_FUSED
    unsigned long long diag_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
    const unsigned long long first_ = last_, first_rt_ = __builtin_amdgcn_s_memrealtime();
#endif
    for (int T = cu; T < n_local; T += n_cu) {
        int qt, st;
        decode(T, qt, st);
        const int q0 = qt * BQP, s0 = st * BS;
        const int nrun = ws_nrun[st];
        const int2 bnd = *reinterpret_cast<const int2*>(ws_bnd + 2 * (size_t)st);
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[j][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        F1 a0, a1;
        F2 b0;
        {   // the tile's first fragments; the barrier behind them frees the buffer for the DMAs of three stages on
            const unsigned base = ring_lds + (unsigned)gi * P::STAGE_BYTES;
            const unsigned aqh = base + lq_h, aql = base + lq_l, ash = base + ls_h, asl = base + ls_l;
            NW_LDS_RD128(a0.bh[0], aqh, 0); NW_LDS_RD128(a0.bh[1], aqh, BLK);
            NW_LDS_RD128(b0.bl[0], aql, 0); NW_LDS_RD128(b0.bl[1], aql, BLK);
            NW_LDS_RD128(a0.al[0], asl, 0 * BLK); NW_LDS_RD128(a0.al[1], asl, 1 * BLK); NW_LDS_RD128(a0.al[2], asl, 2 * BLK);
            NW_LDS_RD128(a0.al[3], asl, 3 * BLK); NW_LDS_RD128(a0.al[4], asl, 4 * BLK); NW_LDS_RD128(a0.al[5], asl, 5 * BLK);
            NW_LDS_RD128(a0.al[6], asl, 6 * BLK); NW_LDS_RD128(a0.al[7], asl, 7 * BLK);
            NW_LDS_RD128(b0.ah[0], ash, 0 * BLK); NW_LDS_RD128(b0.ah[1], ash, 1 * BLK); NW_LDS_RD128(b0.ah[2], ash, 2 * BLK);
            NW_LDS_RD128(b0.ah[3], ash, 3 * BLK); NW_LDS_RD128(b0.ah[4], ash, 4 * BLK); NW_LDS_RD128(b0.ah[5], ash, 5 * BLK);
            NW_LDS_RD128(b0.ah[6], ash, 6 * BLK); NW_LDS_RD128(b0.ah[7], ash, 7 * BLK);
#ifdef NW_ABL_NORD
            a1 = a0;
#endif
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(a0.bh[0]), "+v"(a0.bh[1]), "+v"(a0.al[0]), "+v"(a0.al[1]), "+v"(a0.al[2]), "+v"(a0.al[3]), "+v"(a0.al[4]),
                           "+v"(a0.al[5]), "+v"(a0.al[6]), "+v"(a0.al[7]), "+v"(b0.bl[0]), "+v"(b0.bl[1]), "+v"(b0.ah[0]), "+v"(b0.ah[1]),
                           "+v"(b0.ah[2]), "+v"(b0.ah[3]), "+v"(b0.ah[4]), "+v"(b0.ah[5]), "+v"(b0.ah[6]), "+v"(b0.ah[7]));
            pin();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            pin();
        }
        int kt = 0, gk = 1;         // gk = (kt + 1) mod NB, relative to gi
        auto next_base = [&]() {
            int b = gi + gk;
            b -= (b >= NB) ? NB : 0;
            gk = (gk + 1 == NB) ? 0 : gk + 1;
            return ring_lds + (unsigned)b * P::STAGE_BYTES;
        };
        for (; kt + 2 < nk; kt += 2) {
            run_stage(a0, a1, b0, next_base(), Yes{}, true);
            run_stage(a1, a0, b0, next_base(), Yes{}, true);
        }
        if (kt + 2 == nk) {
            run_stage(a0, a1, b0, next_base(), Yes{}, true);
            run_stage(a1, a0, b0, 0u, No{}, true);
        } else {
            run_stage(a0, a1, b0, 0u, No{}, true);
        }
        gi += nk % NB;
        gi -= (gi >= NB) ? NB : 0;
        NW_PSTAMP(0);
        epilogue_p<RS, KIND, QB, P::NWV>(acc, hdr0 + par * P::HDR_F, nrun, bnd, logit_scale, ws_m, ws_den, ws_num, B, N, q0, s0, st,
                                         wave, lane
#ifdef NW_DIAG_FUSED
                                         , diag_, last_
#endif
                                         );
        par ^= 1;
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

}  // namespace
}  // namespace nw
