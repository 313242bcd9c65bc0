// tile_core.h -- the shared main loop of the MFMA kernels (gfx950 / MI355X only).
//
// One 512-thread workgroup (8 waves) computes the 64 x (16*RS) tile of dot products between 64
// query rows and 16*RS support rows, K streamed in 32-float stages through a three-buffer LDS ring,
// together with the squared row norms of both operands.  Roles are fixed per wave:
//
//   waves 0-3  CONSUMERS  wave w owns query columns [16w, 16w+16) and all RS support blocks:
//                  acc[rs][r] = dot(support 16*rs + 4*(lane>>4) + r, query 16*w + (lane&15))
//              v_mfma_f32_16x16x4_f32 operand map: lane l supplies A[row l&15][k l>>4],
//              B[k l>>4][col l&15].  The k order inside a dot product is free, so lane group
//              g = l>>4 takes the four consecutive floats k = 16t + 4g .. +3 of its row with one
//              ds_read_b128 and feeds four MFMAs.  Fragments are double-buffered in registers: the
//              reads of the next k16-step (and, across the barrier, of the next stage) are in
//              flight while the current step's 4*RS MFMAs issue, so the stream a consumer executes
//              is ds_read_b128 + MFMA + one barrier per stage and nothing else.
//   waves 4-7  LOADERS    stream the stage tiles HBM/L2 -> registers -> LDS two to three stages
//              ahead of the consumers, and accumulate the squared norms from the very
//              registers they copy: no row is read twice, no wave repeats another's work, and the
//              global-load / ds_write issue time never sits in front of an MFMA.
//
// Measured on MI355X (tools/bench_tile.hip, B=256 N=10000 d=512): with loads, LDS writes and norms
// issued by the MFMA waves themselves the loop ran 65 k cycles per workgroup against 41 k of pure
// MFMA issue; the same loop without them 48 k.
//
// LDS rows are 128 B = eight 16-byte slots; ds_read_b128 is serviced per 16-lane group over a 256-B
// bank row, so rows r and r+2 would collide slot for slot: slot ^= (row >> 1) & 7 (conflict-free,
// SQ_LDS_BANK_CONFLICT = 0 measured).
#pragma once
#include <type_traits>
#include "nw_internal.h"

namespace nw {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BQ = 64;          // queries per workgroup
constexpr int BK = 32;          // floats of k per stage: one 128-byte line per row
constexpr int ROW_F4 = BK / 4;  // 16-byte slots per LDS row
constexpr int NCONS = 4;        // consumer (MFMA) waves
constexpr int NLOAD = 4;        // loader waves (one per SIMD, beside one consumer each)
#ifndef NW_NGRP
#define NW_NGRP 4
#endif
constexpr int NGRP = NW_NGRP;   // loader groups: group g owns the stages kt % NGRP == g, so NGRP
                                // stages (NGRP x 28 KB at RS = 10) are in flight per workgroup
constexpr int TILE_THREADS = 64 * (NCONS + NLOAD);
constexpr int LOADER_THREADS = 64 * NLOAD / NGRP;  // threads that copy one stage

template <int RS>
struct TileCfg {
    static constexpr int BS = 16 * RS;
    static constexpr int Q_IT = BQ * ROW_F4 / LOADER_THREADS;  // 16-byte chunks per loader thread
    static constexpr int S_IT = BS * ROW_F4 / LOADER_THREADS;
    static constexpr int TILE_F4 = (BQ + BS) * ROW_F4;         // float4 per stage buffer
    static constexpr size_t STAGE_BYTES = (size_t)3 * TILE_F4 * 16;  // three stage buffers
    static_assert(BQ * ROW_F4 % LOADER_THREADS == 0 && BS * ROW_F4 % LOADER_THREADS == 0, "tile/loader split");
    static_assert(3 * TILE_F4 * 16 <= 160 * 1024, "stage ring exceeds LDS");
};

// Workgroup barrier that publishes this wave's LDS writes but does NOT drain its global loads:
// __syncthreads() fences with s_waitcnt vmcnt(0), which would park a loader wave (and with it every
// consumer waiting at the same barrier) for the full latency of the loads it has just issued.
// The sched_barrier(0) pair pins the instruction order around it: without it hipcc moves the (register-
// only) MFMAs of a stage BELOW the barrier and the wait, i.e. behind "wait for every LDS read in
// flight", which exposes the full LDS latency once per stage (measured: 49.7 k -> see DESIGN.md).
__device__ __forceinline__ void tile_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// The same barrier for a wave whose LDS reads of the buffer being released have all been CONSUMED (an MFMA
// that uses them has been issued in front of the barrier) and whose reads still in flight belong to another
// buffer: nothing to wait for (fused_f16p.h, hand-ordered stage).
__device__ __forceinline__ void tile_barrier_nowait() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }
__device__ __forceinline__ float dot4(const float4 v) { return v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }

// Runs the K loop.  Must be called by all 512 threads.  On return acc[] (consumer waves only) holds
// the dot products and, when NEED_NORM, qn2[0..63] / sn2[0..BS) (LDS, caller-provided, outside the
// staging area) hold the squared norms, visible to every wave (a barrier has been passed); the stage
// buffers are dead.
template <int RS, bool NEED_NORM>
__device__ __forceinline__ void tile_dots(const float* __restrict__ q, const float* __restrict__ s,
                                          int B, int N, int d, int q0, int s0, float4* stage,
                                          float* qn2, float* sn2, f32x4 (&acc)[RS], int rot = 0,
                                          unsigned long long* diag = nullptr) {
    using Cfg = TileCfg<RS>;
#ifdef NW_DIAG_PHASES
    unsigned long long dg[4] = {0, 0, 0, 0};
#define NW_STAMP(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#define NW_ACC(i, a, b) dg[i] += (b) - (a)
#else
#define NW_STAMP(x)
#define NW_ACC(i, a, b)
#endif
    constexpr int Q_IT = Cfg::Q_IT, S_IT = Cfg::S_IT, TILE_F4 = Cfg::TILE_F4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branch
    const int nfull = d / BK;          // full stages
    const int nk = (d + BK - 1) / BK;  // + one partial stage when d % 32 != 0

    if (wave >= NCONS) {
        // ================================ LOADER ================================
        // Each loader shares its SIMD with one MFMA wave, and issue arbitration is "priority, then
        // age": at equal priority the older MFMA wave wins every slot it can use and the loader's
        // ~60 instructions per stage took ~2900 cycles to issue (measured, tools/bench_tile.hip).
        // With the loaders on top they take what little they need and the MFMA waves keep the rest.
#ifndef NW_ABL_NOPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        constexpr int RSTEP = LOADER_THREADS / ROW_F4;  // rows covered by one pass of a loader group
        const int grp = (tid - 64 * NCONS) / LOADER_THREADS;
        const int lt = (tid - 64 * NCONS) % LOADER_THREADS;
        const int crow = lt >> 3, cslot = lt & 7;       // chunk c = lt + LOADER_THREADS*it -> row crow + RSTEP*it
        const float* qsrc[Q_IT];
        const float* ssrc[S_IT];
#pragma unroll
        for (int it = 0; it < Q_IT; ++it) qsrc[it] = q + (size_t)min(q0 + crow + RSTEP * it, B - 1) * d + cslot * 4;
#pragma unroll
        for (int it = 0; it < S_IT; ++it) ssrc[it] = s + (size_t)min(s0 + crow + RSTEP * it, N - 1) * d + cslot * 4;
        float nq[Q_IT], ns[S_IT];
#pragma unroll
        for (int it = 0; it < Q_IT; ++it) nq[it] = 0.f;
#pragma unroll
        for (int it = 0; it < S_IT; ++it) ns[it] = 0.f;

        struct Regs {
            float4 q[Q_IT];
            float4 s[S_IT];
        };
        // The last stage may be partial (d % 32 != 0): its loads read from a clamped in-row address
        // and the store zeroes the dead chunks, so that ONE unconditional load sequence serves every
        // stage (a second, conditional load path makes hipcc merge the two register sets with copies
        // behind an immediate s_waitcnt, which serialises the loader on the full memory latency).
        // K is walked in a per-tile rotated order (chunk (kt + rot) % nk at stage kt; a dot product
        // does not care): with d = 512 every row is 2 KB apart, so all the workgroups of the chip
        // reading "chunk kt of their rows" at the same moment hammer the same one or two L2/memory
        // channels; the rotation spreads simultaneous requests over all of them.
        auto chunk_of = [&](int kt) {
            const int kc = kt + rot;
            return kc >= nk ? kc - nk : kc;
        };
        auto gload = [&](Regs& R, int kt) {
            const int koff = min(chunk_of(kt) * BK, d - 4 - cslot * 4);
#pragma unroll
            for (int it = 0; it < Q_IT; ++it) R.q[it] = *reinterpret_cast<const float4*>(qsrc[it] + koff);
#pragma unroll
            for (int it = 0; it < S_IT; ++it) R.s[it] = *reinterpret_cast<const float4*>(ssrc[it] + koff);
        };
        auto lstore = [&](const Regs& R, int kt) {
            float4* Qs = stage + (kt % 3) * TILE_F4;
            float4* Ss = Qs + BQ * ROW_F4;
            const int kc = chunk_of(kt);
            const float live = (kc * BK + cslot * 4 < d) ? 1.f : 0.f;
#pragma unroll
            for (int it = 0; it < Q_IT; ++it) {
                const int row = crow + RSTEP * it;
                float4 v = R.q[it];
                if (kc >= nfull) v = make_float4(v.x * live, v.y * live, v.z * live, v.w * live);
                Qs[row * ROW_F4 + swz(row, cslot)] = v;
                if (NEED_NORM) nq[it] += dot4(v);
            }
#pragma unroll
            for (int it = 0; it < S_IT; ++it) {
                const int row = crow + RSTEP * it;
                float4 v = R.s[it];
                if (kc >= nfull) v = make_float4(v.x * live, v.y * live, v.z * live, v.w * live);
                Ss[row * ROW_F4 + swz(row, cslot)] = v;
                if (NEED_NORM) ns[it] += dot4(v);
            }
        };

        // Each group keeps ONE register set, used in place: when its turn comes it first retires
        // the stage it holds (requested NGRP iterations earlier) into LDS and then re-issues the
        // loads of its next stage into the same registers.  Latency budget of a load: NGRP stage
        // times; bytes in flight per workgroup: NGRP stage tiles (Little's law: one 28 KB tile in
        // flight per CU at ~2 us loaded latency is only 14 GB/s per CU, half of what the MFMAs eat).
        Regs R;
#pragma unroll
        for (int k0 = 0; k0 < 2; ++k0)
            if (k0 % NGRP == grp && k0 < nk) {
                gload(R, k0);
                lstore(R, k0);
            }
        {
            int kfirst = 2 + ((grp - 2) % NGRP + NGRP) % NGRP;  // first stage >= 2 this group owns
            if (kfirst < nk) gload(R, kfirst);
        }
        tile_barrier();
        // iteration kt retires stage kt+2 into buffer (kt+2) % 3, which the consumers last read in
        // iteration kt-1 (before the barrier that ended it) and read again from iteration kt+1 on.
        for (int kt = 0; kt < nk; ++kt) {
            const int ks = kt + 2;
            NW_STAMP(t0);
            if (ks % NGRP == grp) {
#ifndef NW_ABL_NOLSTORE
                if (ks < nk) lstore(R, ks);
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                NW_STAMP(t1);
                NW_ACC(0, t0, t1);
#ifndef NW_ABL_NOGLOAD
                if (ks + NGRP < nk) gload(R, ks + NGRP);
#endif
                NW_STAMP(t2);
                NW_ACC(1, t1, t2);
            }
            NW_STAMP(t3);
            tile_barrier();
            NW_STAMP(t4);
            NW_ACC(2, t3, t4);
        }
#ifdef NW_DIAG_PHASES
        if (diag && tid == 64 * NCONS) { diag[4 * blockIdx.x + 0] = dg[0]; diag[4 * blockIdx.x + 1] = dg[1]; diag[4 * blockIdx.x + 2] = dg[2]; }
#endif

        if (NEED_NORM) {
            // the 8 threads lt&7 = 0..7 of one row hold the 8 slot-partials of its squared norm; every loader
            // group parks its partial in the (dead) stage ring, the groups are added in order below: no float
            // atomics, the norms are bit-reproducible
            float* part = reinterpret_cast<float*>(stage) + grp * (BQ + Cfg::BS);
#pragma unroll
            for (int it = 0; it < Q_IT; ++it) {
                float v = nq[it];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                if (cslot == 0) part[crow + RSTEP * it] = v;
            }
#pragma unroll
            for (int it = 0; it < S_IT; ++it) {
                float v = ns[it];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                if (cslot == 0) part[BQ + crow + RSTEP * it] = v;
            }
        }
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
        const int rsw = (i >> 1) & 7;  // == ((16*rs + i) >> 1) & 7 for every rs, and for qrow
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
        };

#pragma unroll
        for (int r = 0; r < RS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_barrier();  // stages 0 and 1 are in LDS
        Frag f0, f1;
        int b0 = 0, b1 = 1, b2 = 2;  // buffers of stage kt, kt+1, kt+2
        if (nk > 0) load_frags(f0, 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            // sched_barrier(0): keep the fragment reads of the NEXT step in front of this step's MFMAs
            // (hipcc otherwise sinks them behind most of the MFMAs they are meant to hide under)
            load_frags(f1, b0, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) load_frags(f0, b1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f1);
            NW_STAMP(c0);
            tile_barrier();
            NW_STAMP(c1);
            NW_ACC(3, c0, c1);
            const int tb = b0;
            b0 = b1;
            b1 = b2;
            b2 = tb;
        }
    }
#ifdef NW_DIAG_PHASES
    if (diag && tid == 0) diag[4 * blockIdx.x + 3] = dg[3];
#endif
    if (NEED_NORM) {
        __syncthreads();  // the loader groups' partial norms are in the stage ring (its last stage has been consumed)
        const float* part = reinterpret_cast<const float*>(stage);
        for (int x = tid; x < BQ + Cfg::BS; x += TILE_THREADS) {
            float v = part[x];
#pragma unroll
            for (int gi = 1; gi < NGRP; ++gi) v += part[gi * (BQ + Cfg::BS) + x];
            if (x < BQ) qn2[x] = v; else sn2[x - BQ] = v;
        }
        __syncthreads();  // norms published
    }
}

// XCD-aware block decode: the n_qtiles workgroups that stream the same support tile get block ids
// that are equal mod 8, i.e. land on the same XCD / L2 under round-robin dispatch (speed only; any
// placement is correct).  Returns false for the padding blocks of the last group.
__device__ __forceinline__ bool decode_block(int n_stiles, int n_qtiles, int& qt, int& st) {
    const int per_grp = 8 * n_qtiles;
    const int grp = blockIdx.x / per_grp, rem = blockIdx.x % per_grp;
    qt = rem >> 3;
    st = grp * 8 + (rem & 7);
    return st < n_stiles;
}
inline int padded_grid(int n_stiles, int n_qtiles) { return ((n_stiles + 7) / 8) * 8 * n_qtiles; }

}  // namespace nw
