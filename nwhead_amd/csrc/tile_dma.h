// tile_dma.h -- main loop of the MFMA kernels, LDS-DMA edition (gfx950 / MI355X only).
//
// Same tile and the same consumer stream as tile_core.h (64 queries x 16*RS supports per 512-thread
// workgroup, waves 0-3 = MFMA consumers, waves 4-7 = loaders), but the loaders move the stage tiles
// with global_load_lds_dwordx4 (HBM/L2 -> LDS directly, no VGPR data, no ds_write) into a FOUR-buffer
// ring, two to three stages ahead of the consumers:
//
//   loader wave, iteration kt :  issue the 2 + RS/2 DMAs of stage kt+3 (1 KB each: 8 rows x 128 B)
//                                s_waitcnt vmcnt(2 + RS/2)   -> stage kt+2 has landed
//                                barrier                      -> ... and is published
//   consumer wave, iteration kt: reads stage kt (and prefetches the first fragments of stage kt+1),
//                                MFMAs, barrier.
//
// A loader executes ~10 instructions per stage, so it hardly competes with the MFMA wave it shares a
// SIMD with (the register-staged loader of tile_core.h needs ~70).  That matters because on gfx950
// the fp32 MFMA runs at exactly the fp32 VALU rate and, as measured here, does NOT overlap with VALU
// work on the same SIMD: every v_fma issued by either wave of a SIMD costs ~10 cycles of MFMA time
// (88 norm FMAs per stage in the consumers = +14 k cycles on a 41 k-cycle loop).  So the consumer
// stream is kept to ds_read_b128 + MFMA, and the row norms are either
//   * skipped for the supports when the caller passes precomputed norms (the resident bank of
//     'full' inference: NWNet.precompute() caches them), or
//   * accumulated by the consumers from their fragments (generic path, ~25 % slower: measured 66 k
//     vs 51 k cycles per workgroup at B=256 N=10000 d=512; splitting the work over the waves by
//     stage or by block did not help with hipcc's code for it -- a hand-scheduled version is open).
//
// The LDS image is the same XOR-swizzled [row][8 x 16 B] layout; an LDS-DMA writes lane l of a wave
// at M0 + 16*l, i.e. linearly, so the swizzle is applied to the per-lane SOURCE address instead.
// Requires d % 32 == 0 (a DMA cannot zero-fill a partial stage); other d use tile_core.h.
#pragma once
#include "tile_core.h"

namespace nw {

// Ring depth.  Tiles of up to 5 support blocks keep a workgroup under 80 KB of LDS and 128 VGPRs,
// so TWO workgroups share a CU (4 waves per SIMD): one's barrier bubbles, DMA prologue and VALU
// epilogue run under the other's MFMAs.
template <int RS>
struct DmaCfg {
    static constexpr int BS = 16 * RS;
    static constexpr int TILE_F4 = (BQ + BS) * ROW_F4;
    static constexpr int NT = (BQ + BS) / 8;                 // DMA instructions per stage (1 KB each)
    static constexpr int NI = (NT + NLOAD - 1) / NLOAD;      // ... per loader wave (waves lw < NT % NLOAD, or all)
    static constexpr int NI_LO = NT / NLOAD;                 // ... for the other loader waves
    // Ring depth: as many stage buffers as the CU's LDS holds next to the header (one workgroup per CU
    // for the tall tiles, two for RS <= 5), at most 8: NBUF-1 stages are in flight per loader wave, and at
    // small grids (one tile per CU, e.g. T) the loop is bound by bytes in flight / fill latency.
    static constexpr size_t STAGE_ONE = (size_t)TILE_F4 * 16;
    static constexpr size_t LDS_BUDGET = ((RS <= 5) ? 80 : 160) * 1024 - 4096;  // 4 KB: header of the fused kernels
    static constexpr int NBUF_FIT = (int)(LDS_BUDGET / STAGE_ONE);
#ifdef NW_NBUF
    static constexpr int NBUF = NW_NBUF;
#else
    static constexpr int NBUF = NBUF_FIT > 8 ? 8 : (NBUF_FIT < 4 ? 4 : NBUF_FIT);
#endif
    static constexpr size_t STAGE_BYTES = (size_t)NBUF * STAGE_ONE;
    static constexpr int WAVES_PER_SIMD = (RS <= 5) ? 4 : 2; // launch bound: 128 or 256 VGPRs
    static_assert((NBUF - 3) * NI < 64, "vmcnt is a 6-bit field");
};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#ifdef NW_DIAG_FUSED   // diagnostic build only (tools/bench_fused.hip): end of the consumers' query prologue per workgroup
__device__ unsigned long long nw_diag_q[4096];
#endif

// Exponent of the split-row scale: 2^e with rowmax * 2^e in [2^13, 2^14) (e = 0 for an all-zero / non-finite row;
// capped so that 2^e stays finite and 2^-e normal for rows of subnormal magnitude).  The rule of split.hip.
__device__ __forceinline__ int split_exponent(float rowmax) {
    int e = 0;
    if (rowmax > 0.f && rowmax < INFINITY) {
        e = 14 - __builtin_amdgcn_frexp_expf(rowmax);  // rowmax = f * 2^x, f in [0.5, 1)
        if (e > 126) e = 126;
    }
    return e;
}

// (wait_vmcnt<N>: nw_internal.h)

// The loader role (waves 4-7), shared by the fp32 and the split-fp16 consumers: the stage image in
// LDS is byte-identical in both (128 B per row per stage).
//
// QRAW (split-fp16 consumers, raw fp32 queries; tile_f16.h): the query rows are DMA'd RAW and the consumers split
// their fragment in registers.  They read slots 2g and 2g+1 of a query row (k = 8g .. 8g+7 as fp32) instead of g
// and 4+g, so the query rows' 16-byte slots are swizzled with (row >> 1) & 5 -- conflict-free for that pattern --
// while the support rows keep (row >> 1) & 7.
//   (Splitting in place in LDS by the loader waves was measured and dropped: the loaders are the pole of this
//    loop -- 7 DMA issues of ~100 cycles per stage against 480 cycles of MFMA -- and the conversion's LDS round
//    trip added 390 cycles per stage, 19.3 vs 13.4 us at T; row statistics in the loaders' prologue serialise
//    behind the DMA fill, +4.6 k cycles.)
template <int RS, bool QRAW = false>
__device__ __forceinline__ void dma_loader_run(const float* __restrict__ q, const float* __restrict__ s,
                                               int B, int N, int d, int q0, int s0, float4* stage,
                                               int rot, int wave, int lane) {
    using Cfg = DmaCfg<RS>;
    constexpr int TILE_F4 = Cfg::TILE_F4, NI = Cfg::NI, NI_LO = Cfg::NI_LO, NT = Cfg::NT;
    const int nk = d / BK;
    {
        const int lw = wave - NCONS;
        // leave the K youngest stages of this wave's DMAs in flight
        auto wait_ahead_impl = [](auto kc, bool lng) {
            constexpr int K = decltype(kc)::value;
            if (lng) wait_vmcnt<K * NI>(); else wait_vmcnt<K * NI_LO>();
        };
#define wait_ahead_K(K, lng) wait_ahead_impl(std::integral_constant<int, (K)>{}, (lng))
        // instruction n = lw + NLOAD*m covers stage rows 8n .. 8n+7 (Q rows first, then S rows)
        // source = wave-uniform base (q or s, advanced by the stage's k offset) + a per-lane 32-bit
        // byte offset that never changes: no vector arithmetic per DMA.
        unsigned voff[NI];
#pragma unroll
        for (int m = 0; m < NI; ++m) {
            const int R = 8 * (lw + NLOAD * m) + (lane >> 3);
            // source-side swizzle; QRAW query rows: (row >> 1) & 5, see above
            const int lslot = (lane & 7) ^ ((R >> 1) & ((QRAW && 8 * NLOAD * m < BQ) ? 5 : 7));
            // offsets are relative to the tile's first row (the bases below carry q0 / s0 in 64 bits): banks
            // beyond 4 GB are fine, a tile never spans more than (BQ + BS) * d * 4 bytes
            const int rel = (8 * NLOAD * m < BQ) ? min(q0 + R, B - 1) - q0 : min(s0 + R - BQ, N - 1) - s0;
            voff[m] = ((unsigned)rel * (unsigned)d + lslot * 4) * 4u;
        }
        auto issue = [&](int kt) {
            int kc = kt + rot;
            if (kc >= nk) kc -= nk;
            float4* buf = stage + ((unsigned)kt % Cfg::NBUF) * TILE_F4;
            const char* qb = reinterpret_cast<const char*>(q + (size_t)q0 * d) + (size_t)kc * BK * 4;
            const char* sb = reinterpret_cast<const char*>(s + (size_t)s0 * d) + (size_t)kc * BK * 4;
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                if (NI != NI_LO && m == NI - 1 && lw + NLOAD * m >= NT) break;  // uneven split: last slot of the short waves
                const char* g = ((8 * NLOAD * m < BQ) ? qb : sb) + voff[m];  // rows 8n..8n+7 are all Q or all S
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(buf + 64 * (lw + NLOAD * m)),
                                                 16, 0, 0);
            }
        };
        const bool long_wave = (NI == NI_LO) || (lw < NT % NLOAD);
        // prologue: stages 0 .. NBUF-2 in flight; the consumers start once 0 and 1 have landed
        constexpr int AHEAD = Cfg::NBUF - 1;
#pragma unroll
        for (int k0 = 0; k0 < AHEAD; ++k0)
            if (k0 < nk) issue(k0);
        if (nk >= AHEAD) wait_ahead_K(AHEAD - 2, long_wave); else wait_vmcnt<0>();
        tile_barrier();
        // iteration kt: issue stage kt+AHEAD into the buffer the consumers left at the last barrier,
        // then wait until stage kt+2 has landed (all but the AHEAD-2 youngest stages of this wave)
        for (int kt = 0; kt < nk; ++kt) {
#ifndef NW_ABL_NODMA
            if (kt + AHEAD < nk) {
                issue(kt + AHEAD);
                wait_ahead_K(AHEAD - 2, long_wave);
            } else {
                wait_vmcnt<0>();
            }
#endif
#ifndef NW_ABL_NOBAR
            tile_barrier();
#endif
        }
    }
}


// Must be called by all 512 threads; d % 32 == 0, d >= 32.
// NEED_QN / NEED_SN: accumulate squared norms of the query / support rows into qn2[0..63] / sn2[0..BS)
// (LDS, outside the stage ring).  On return a barrier has been passed and the ring is dead.
template <int RS, bool NEED_QN, bool NEED_SN>
__device__ __forceinline__ void tile_dots_dma(const float* __restrict__ q, const float* __restrict__ s,
                                              int B, int N, int d, int q0, int s0, float4* stage,
                                              float* qn2, float* sn2, f32x4 (&acc)[RS], int rot) {
    using Cfg = DmaCfg<RS>;
    constexpr int TILE_F4 = Cfg::TILE_F4, NI = Cfg::NI, NI_LO = Cfg::NI_LO, NT = Cfg::NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = d / BK;

    if (wave >= NCONS) {
        dma_loader_run<RS>(q, s, B, N, d, q0, s0, stage, rot, wave, lane);
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};  // loaders hold no results
    } else {
        // ================================ CONSUMER ================================
        const int i = lane & 15, g = lane >> 4;
        struct Frag {
            float4 b;
            float4 a[RS];
        };
        const int qrow = 16 * wave + i;
        const int rsw = (i >> 1) & 7;
        float4 sq4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float sqs[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            sqs[r] = 0.f;
        }
        auto load_frags = [&](Frag& f, int buf, int t) {
            const float4* Qs = stage + buf * TILE_F4;
            const float4* Ss = Qs + BQ * ROW_F4;
            const int slot = (4 * t + g) ^ rsw;
            f.b = Qs[qrow * ROW_F4 + slot];
#pragma unroll
            for (int r = 0; r < RS; ++r) f.a[r] = Ss[(16 * r + i) * ROW_F4 + slot];
        };
        auto mfma_step = [&](const Frag& f) {
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[r].x, f.b.x, acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[r].y, f.b.y, acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[r].z, f.b.z, acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[r].w, f.b.w, acc[r], 0, 0, 0);
            if (NEED_QN) {  // a wave's 16 query rows are its own; four independent FMAs, no dependent chain
                sq4.x = __builtin_fmaf(f.b.x, f.b.x, sq4.x);
                sq4.y = __builtin_fmaf(f.b.y, f.b.y, sq4.y);
                sq4.z = __builtin_fmaf(f.b.z, f.b.z, sq4.z);
                sq4.w = __builtin_fmaf(f.b.w, f.b.w, sq4.w);
            }
            if (NEED_SN) {                  // generic path (no cached support norms): every wave, every block
#pragma unroll
                for (int r = 0; r < RS; ++r) sqs[r] += dot4(f.a[r]);
            }
        };

        tile_barrier();  // stages 0 and 1 have landed
        Frag f0, f1;
        load_frags(f0, 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int b0 = (unsigned)kt % Cfg::NBUF, b1 = (unsigned)(kt + 1) % Cfg::NBUF;
            // sched_barrier(0): keep the fragment reads of the NEXT step in front of this step's MFMAs
            // (hipcc otherwise sinks them behind most of the MFMAs they are meant to hide under)
            load_frags(f1, b0, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) load_frags(f0, b1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f1);
#ifndef NW_ABL_NOBAR
            tile_barrier();
#endif
        }
        // lanes i, i+16, i+32, i+48 hold the four k-slices of one row's squared norm
        if (NEED_QN) {
            float sqq = (sq4.x + sq4.y) + (sq4.z + sq4.w);
            sqq += __shfl_xor(sqq, 16);
            sqq += __shfl_xor(sqq, 32);
            if (g == 0) qn2[qrow] = sqq;
        }
        if (NEED_SN) {
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                float v = sqs[r];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if ((r & (NCONS - 1)) == wave && g == 0) sn2[16 * r + i] = v;  // all four waves hold the same value
            }
        }
    }
    if (NEED_QN || NEED_SN) __syncthreads();
}

}  // namespace nw
