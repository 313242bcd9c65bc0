// tile_f16.h -- main loop on the fp16 matrix cores with fp32-grade accuracy (gfx950 / MI355X only).
//
// fp32 MFMA runs at 1/16 of the fp16/bf16 MFMA rate.  Every fp32 operand x (scaled by a power of two
// per row, see below) is split once into two fp16 numbers, x = h + l + O(2^-23 |x|), h = fp16(x),
// l = fp16(x - h), and a dot product is evaluated as
//        sum_k a_k b_k  ~=  sum_k ( ah_k bh_k + ah_k bl_k + al_k bh_k )        (al*bl ~ 2^-22 dropped)
// i.e. THREE v_mfma_f32_16x16x32_f16 (16 cycles each) per 32 k instead of EIGHT v_mfma_f32_16x16x4_f32
// (32 cycles each): 5.3x less matrix-pipe time; products are exact and accumulated in fp32.  Error of
// a dot product ~2.4e-7 * sum|a_k b_k|, the same order as an fp32 FMA chain (~1e-7 * sum|a_k b_k|) and
// two orders below the 1e-5 parity bar.
//
// Operand format ("split rows", written by nw_split_rows_f16x2): a (rows,d) array with the byte size
// and row stride of the fp32 original, in which every 128-byte chunk of a row holds [32 x h | 32 x l]
// (fp16) for the same 32 k.  So the loaders, the LDS image (128 B per row per stage, XOR-swizzled
// 16-byte slots) and the bank footprint are exactly those of the fp32 path (tile_dma.h); only the
// consumers read differently: lane (i,g) takes slot g (eight h values, k = 8g..8g+7) and slot 4+g
// (the matching l values) of its row.  Rows are pre-scaled by 2^e (e per row, rowmax -> [2^13, 2^14))
// so that both halves stay in fp16's normal range whatever the magnitude of the features; the caller
// undoes the scale in the epilogue (dot = acc * 2^-(e_q + e_s)).
#pragma once
#include "tile_dma.h"

namespace nw {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// Must be called by all 512 threads; d % 32 == 0, d >= 32; q and s are SPLIT rows (see above).
template <int RS>
__device__ __forceinline__ void tile_dots_f16x2(const float* __restrict__ q, const float* __restrict__ s,
                                                int B, int N, int d, int q0, int s0, float4* stage,
                                                f32x4 (&acc)[RS], int rot) {
    using Cfg = DmaCfg<RS>;
    constexpr int TILE_F4 = Cfg::TILE_F4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = d / BK;

    if (wave >= NCONS) {
        dma_loader_run<RS>(q, s, B, N, d, q0, s0, stage, rot, wave, lane);
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};  // loaders hold no results
    } else {
        const int i = lane & 15, g = lane >> 4;
        struct Frag {
            float4 bh, bl;
            float4 ah[RS], al[RS];
        };
        const int qrow = 16 * wave + i;
        const int rsw = (i >> 1) & 7;
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto load_frags = [&](Frag& f, int buf) {
            const float4* Qs = stage + buf * TILE_F4;
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
        auto mfma_stage = [&](const Frag& f) {
            // small terms first, the dominant h*h product last
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = mm(f.al[r], f.bh, acc[r]);
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = mm(f.ah[r], f.bl, acc[r]);
#pragma unroll
            for (int r = 0; r < RS; ++r) acc[r] = mm(f.ah[r], f.bh, acc[r]);
        };

        tile_barrier();  // stages 0 and 1 have landed
        if (RS > 5) {
            // one workgroup per CU: fragments double-buffered in registers.  The 2(RS+1) reads of the
            // next stage are interleaved one-to-one with the first MFMAs of the current stage
            // (sched_group_barrier): a ds_read_b128 issued between two 16-cycle MFMAs is free, issued
            // in a block in front of them it is ~8 cycles of idle matrix pipe each.
            auto interleave = [&]() {
#pragma unroll
                for (int x = 0; x < 2 * (RS + 1); ++x) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * RS - 2 * (RS + 1), 0);  // the remaining MFMAs
            };
            Frag f0, f1;
            load_frags(f0, 0);
            int kt = 0;
            for (; kt + 2 < nk; kt += 2) {
                load_frags(f1, (unsigned)(kt + 1) % Cfg::NBUF);
                mfma_stage(f0);
                interleave();
                tile_barrier();
                load_frags(f0, (unsigned)(kt + 2) % Cfg::NBUF);
                mfma_stage(f1);
                interleave();
                tile_barrier();
            }
            for (; kt < nk; ++kt) {  // tail: one or two stages, nothing further to prefetch
                if (kt + 1 < nk) load_frags(f1, (unsigned)(kt + 1) % Cfg::NBUF);
                mfma_stage(f0);
                tile_barrier();
                f0 = f1;
            }
        } else {
            // two workgroups per CU (128 VGPRs): one fragment set; the other three waves of the SIMD
            // cover the LDS latency
            Frag f;
            for (int kt = 0; kt < nk; ++kt) {
                load_frags(f, (unsigned)kt % Cfg::NBUF);
                mfma_stage(f);
                tile_barrier();
            }
        }
    }
}

}  // namespace nw
