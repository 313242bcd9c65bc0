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

// Must be called by all 512 threads; d % 32 == 0, d >= 32; s holds SPLIT rows (see above).
// QRAW = false: q holds split rows too (nw_split_rows_kernel ran over the query batch).
// QRAW = true : q holds the caller's RAW fp32 rows.  While the loaders fill the DMA pipeline, each consumer wave
//               reads its own 16 query rows once from global memory (four lanes per row; L2-resident: every support
//               tile's workgroup reads the same 64 rows) for the row maximum and squared norm -> 2^-e and the norm go
//               to qsc_s / qn2_s (LDS header) for the epilogue; in the loop a lane splits its query fragment in
//               registers, one stage ahead of its use (8 values per stage: scale, h = fp16(x), l = fp16(x - h);
//               24 packed VALU ops under 3*RS MFMAs).  No launch in front of the tile kernel: 16.4 us at T against
//               13.4 + 4.4 us (the split launch) + a kernel boundary.
//               Measured on the way (T, cycles per workgroup, tools/bench_fused.hip): raw layout alone +0; the
//               split in the loop +2.0 k; the statistics pass +3.4 k (its 128 KB per workgroup compete with the DMA
//               fill).  Dropped: statistics by the loader waves (serialise behind their DMA issue, +4.6 k); the
//               split by the loader waves in place in LDS (they are the pole of the loop: 7 DMA issues of ~100
//               cycles per stage; +390 cycles per stage); finding the scale on the way with an exact rescale of
//               the accumulators when a stage outgrows it (no pre-pass, but the branch at the stage boundary cost
//               more than the pass it saved: 17.8 k / 24.4 k cycles against 18.2 k).
template <int RS, bool QRAW = false>
__device__ __forceinline__ void tile_dots_f16x2(const float* __restrict__ q, const float* __restrict__ s,
                                                int B, int N, int d, int q0, int s0, float4* stage,
                                                f32x4 (&acc)[RS], int rot, float* qn2_s = nullptr,
                                                float* qsc_s = nullptr) {
    using Cfg = DmaCfg<RS>;
    constexpr int TILE_F4 = Cfg::TILE_F4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = d / BK;

    if (wave >= NCONS) {
        dma_loader_run<RS, QRAW>(q, s, B, N, d, q0, s0, stage, rot, wave, lane);
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
        const int qsw = (i >> 1) & 5;  // QRAW: swizzle of the raw query rows (tile_dma.h)
#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        float up = 1.f;  // QRAW: 2^e of this lane's query row
        if (QRAW) {
            // lanes 4r .. 4r+3 take row r of this wave's 16: a quarter of the row each (chunks j, j+4, ...), 16 loads
            // in flight, two shuffles for the row's maximum and squared norm
            const int qr = lane >> 2, qj = lane & 3;
            const float4* src = reinterpret_cast<const float4*>(q + (size_t)min(q0 + 16 * wave + qr, B - 1) * d);
            const int n16 = d >> 4;  // float4 chunks per lane
            float mx = 0.f, n2 = 0.f;
            auto take = [&](const float4 v) {
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                n2 += dot4(v);
            };
            int t = 0;
            for (; t + 16 <= n16; t += 16) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = src[qj + 4 * (t + u)];
#pragma unroll
                for (int u = 0; u < 16; ++u) take(v[u]);
            }
            for (; t < n16; ++t) take(src[qj + 4 * t]);
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            n2 += __shfl_xor(n2, 1);
            n2 += __shfl_xor(n2, 2);
            const int e = split_exponent(mx);
            if (qj == 0) {
                qn2_s[16 * wave + qr] = n2;
                qsc_s[16 * wave + qr] = __builtin_ldexpf(1.f, -e);
            }
            // a wave's LDS operations execute in order: the row this lane multiplies with is one it has just written
            up = __builtin_ldexpf(1.f, 1 - __builtin_amdgcn_frexp_expf(qsc_s[qrow]));  // qsc_s = 2^-e exactly
#ifdef NW_DIAG_FUSED
            if (tid == 0) nw_diag_q[blockIdx.x & 4095] = __builtin_amdgcn_s_memtime();
#endif
        }
        // split this lane's raw query fragment (k = 8g .. 8g+7 of its row) into fp16 halves, in place
        auto split_q = [&](Frag& f) {
#ifdef NW_ABL_NOCVT   // timing ablation only (tools/bench_fused.hip): wrong results
            return;
#endif
            const float x[8] = {f.bh.x * up, f.bh.y * up, f.bh.z * up, f.bh.w * up,
                                f.bl.x * up, f.bl.y * up, f.bl.z * up, f.bl.w * up};
            half8 h, l;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                h[k] = (_Float16)x[k];
                l[k] = (_Float16)(x[k] - (float)h[k]);
            }
            f.bh = __builtin_bit_cast(float4, h);
            f.bl = __builtin_bit_cast(float4, l);
        };
        auto load_frags = [&](Frag& f, int buf) {
            const float4* Qs = stage + buf * TILE_F4;
            const float4* Ss = Qs + BQ * ROW_F4;
            const int sh = g ^ rsw, sl = (4 + g) ^ rsw;
            f.bh = Qs[qrow * ROW_F4 + (QRAW ? ((2 * g) ^ qsw) : sh)];
            f.bl = Qs[qrow * ROW_F4 + (QRAW ? ((2 * g + 1) ^ qsw) : sl)];
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
            if (QRAW) split_q(f0);
            int kt = 0;
            for (; kt + 2 < nk; kt += 2) {
                load_frags(f1, (unsigned)(kt + 1) % Cfg::NBUF);
                mfma_stage(f0);
                if (QRAW) split_q(f1);  // the next stage's query fragment, under this stage's MFMAs
                interleave();
                tile_barrier();
                load_frags(f0, (unsigned)(kt + 2) % Cfg::NBUF);
                mfma_stage(f1);
                if (QRAW) split_q(f0);
                interleave();
                tile_barrier();
            }
            for (; kt < nk; ++kt) {  // tail: one or two stages, nothing further to prefetch
                if (kt + 1 < nk) load_frags(f1, (unsigned)(kt + 1) % Cfg::NBUF);
                mfma_stage(f0);
                if (QRAW && kt + 1 < nk) split_q(f1);
                tile_barrier();
                f0 = f1;
            }
        } else {
            // two workgroups per CU (128 VGPRs): one fragment set; the other three waves of the SIMD
            // cover the LDS latency
            Frag f;
            for (int kt = 0; kt < nk; ++kt) {
                load_frags(f, (unsigned)kt % Cfg::NBUF);
                if (QRAW) split_q(f);
                mfma_stage(f);
                tile_barrier();
            }
        }
    }
}

}  // namespace nw
