// scores.hip -- query x support score matrix for the NW head (gfx950 / MI355X only).
//
// Replaces the reference's similarity modules (nwhead/kernel.py:13-44) as they are called from
// NWHead.forward (nwhead/nw.py:283), NWNet.get_neighbors (nw.py:248) and KNN (utils.py:187).
//
// Two kernels:
//   nw_scores_mfma_kernel    shared 2-D support, N > 25, d % 4 == 0: dot products on the fp32
//                            matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains), row
//                            norms accumulated from the same LDS fragments, score epilogue fused.
//                            This is torch.cdist's "matmul form" regime (N > 25).
//   nw_scores_direct_kernel  everything else (N <= 25: torch's direct-difference regime; 3-D
//                            per-query supports; odd d): one wave per (query, support) pair.
#include <type_traits>
#include "tile_core.h"

namespace nw {
namespace {

// Workgroup tile = 64 queries x (16*RS) supports; main loop in tile_core.h.
template <int RS, int KIND>
__global__ __launch_bounds__(TILE_THREADS) void nw_scores_mfma_kernel(
    const float* __restrict__ q, const float* __restrict__ s, float* __restrict__ scores,
    const float* __restrict__ logit_scale, int B, int N, int d, int n_stiles, int n_qtiles) {
    using Cfg = TileCfg<RS>;
    constexpr int BS = Cfg::BS;
    constexpr bool NEED_NORM = (KIND != NW_SCORE_DOT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qn2 = reinterpret_cast<float*>(smem);   // [64]
    float* sn2 = qn2 + 64;                         // [192]
    float4* stage = reinterpret_cast<float4*>(smem + 1024);

    int qt, st;
    if (!decode_block(n_stiles, n_qtiles, qt, st)) return;
    const int q0 = qt * BQ, s0 = st * BS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;

    f32x4 acc[RS];
    tile_dots<RS, NEED_NORM>(q, s, B, N, d, q0, s0, stage, qn2, sn2, acc);

    float scale = 1.f;
    if (KIND == NW_SCORE_CLIP) scale = expf(*logit_scale);
    if (wave >= NCONS) return;  // loader waves hold no accumulators
    const int qrow = 16 * wave + i;
    const int b = q0 + qrow;
    if (b >= B) return;
    const float qn = NEED_NORM ? qn2[qrow] : 0.f;
    float* orow = scores + (size_t)b * N;
    const bool vec_ok = (N & 3) == 0;
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int j = s0 + 16 * r + 4 * g;
        if (j >= N) continue;
        float4 n2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (NEED_NORM) n2 = *reinterpret_cast<const float4*>(sn2 + 16 * r + 4 * g);
        float4 o;
        o.x = score_from_dot<KIND>(acc[r][0], qn, n2.x, scale);
        o.y = score_from_dot<KIND>(acc[r][1], qn, n2.y, scale);
        o.z = score_from_dot<KIND>(acc[r][2], qn, n2.z, scale);
        o.w = score_from_dot<KIND>(acc[r][3], qn, n2.w, scale);
        if (vec_ok) {  // N % 4 == 0 and j % 4 == 0: the whole quad is in range and 16-B aligned
            *reinterpret_cast<float4*>(orow + j) = o;
        } else {
            orow[j] = o.x;
            if (j + 1 < N) orow[j + 1] = o.y;
            if (j + 2 < N) orow[j + 2] = o.z;
            if (j + 3 < N) orow[j + 3] = o.w;
        }
    }
}

// One wave per (query b, support j): direct-difference form for the distance kernels (torch's
// cdist regime for N <= 25: exact 0 for identical rows), plain dot/norms for the others.
template <int KIND>
__device__ __forceinline__ void direct_pair(const float* __restrict__ x, const float* __restrict__ y,
                                            float* __restrict__ scores, const float* __restrict__ logit_scale,
                                            int64_t pair, int64_t d, int lane) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (KIND == NW_SCORE_EUCLIDEAN) {
        for (int64_t k = lane; k < d; k += 64) {
            const float t = x[k] - y[k];
            a0 += t * t;
        }
        a0 = wave_sum(a0);
        if (lane == 0) scores[pair] = -sqrtf(a0);
        return;
    }
    for (int64_t k = lane; k < d; k += 64) {
        const float xv = x[k], yv = y[k];
        a0 += xv * yv;
        a1 += xv * xv;
        a2 += yv * yv;
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    float out;
    if (KIND == NW_SCORE_HYPERSPHERE) {
        // direct form on the normalised rows: sum_k (x_k/|x| - y_k/|y|)^2
        const float inq = 1.f / fmaxf(sqrtf(a1), NW_NORM_EPS), ins = 1.f / fmaxf(sqrtf(a2), NW_NORM_EPS);
        float acc = 0.f;
        for (int64_t k = lane; k < d; k += 64) {
            const float t = x[k] * inq - y[k] * ins;
            acc += t * t;
        }
        out = -sqrtf(wave_sum(acc));
    } else {
        const float scale = (KIND == NW_SCORE_CLIP) ? expf(*logit_scale) : 1.f;
        out = score_from_dot<KIND>(a0, a1, a2, scale);
    }
    if (lane == 0) scores[pair] = out;
}

// one wave per (query, support) pair: the training episodes' shapes (a few dozen queries, a few dozen supports)
template <int KIND>
__global__ __launch_bounds__(256) void nw_scores_direct_kernel(
    const float* __restrict__ q, const float* __restrict__ s, float* __restrict__ scores,
    const float* __restrict__ logit_scale, int64_t B, int64_t N, int64_t d, int sup_batched) {
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pair >= B * N) return;
    const int64_t b = pair / N, j = pair % N;
    direct_pair<KIND>(q + b * d, s + (sup_batched ? (b * N + j) * d : j * d), scores, logit_scale, pair, d, lane);
}

// one wave per QUERY, its supports in turn (same arithmetic per pair): with thousands of queries a wave per pair re-reads
// every query row N times through L2 (65536 x 20 x 512: 0.64 ms); here the row stays in the wave's L1
template <int KIND>
__global__ __launch_bounds__(256) void nw_scores_direct_rows_kernel(
    const float* __restrict__ q, const float* __restrict__ s, float* __restrict__ scores,
    const float* __restrict__ logit_scale, int64_t B, int64_t N, int64_t d, int sup_batched) {
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    for (int64_t j = 0; j < N; ++j)
        direct_pair<KIND>(q + b * d, s + (sup_batched ? (b * N + j) * d : j * d), scores, logit_scale, b * N + j, d, lane);
}

// Per-query supports (sx of shape (B, N, d): every query has its own support rows, nw/nw.py:266-289 with a 3-D sx) at large
// N: nothing is shared between queries, so the work is one pass over B N d floats -- HBM-bound.  Sixteen lanes per support
// row (float4 loads: a wave reads four rows' 256-byte segments per instruction, sixteen rows = sixteen loads in flight per
// lane group of a wave), the direct-difference form of direct_pair (same arithmetic per pair, different summation order),
// the 16-lane sums by four DPP rotations.  One wave per pair (nw_scores_direct_kernel) ran 16 x 5000 x 512 at 0.37 TB/s.
template <int KIND>
__global__ __launch_bounds__(256) void nw_scores_batched_stream_kernel(
    const float* __restrict__ q, const float* __restrict__ s, float* __restrict__ scores,
    const float* __restrict__ logit_scale, int64_t N, int64_t d) {
    constexpr int ROWS_WAVE = 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int64_t b = blockIdx.y;
    const float* __restrict__ x = q + b * d;
    const float* __restrict__ sb = s + b * N * d;
    float* __restrict__ orow = scores + b * N;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * ROWS_WAVE;
    auto rowsum = [](float v) {
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
        return v;
    };
    const float scale = (KIND == NW_SCORE_CLIP) ? expf(*logit_scale) : 1.f;
#pragma unroll 1
    for (int it = 0; it < ROWS_WAVE / 4; ++it) {
        const int64_t j = row0 + 4 * it + g;
        if (row0 + 4 * it >= N) break;                       // wave-uniform
        const float* __restrict__ y = sb + (j < N ? j : N - 1) * d;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        if (KIND == NW_SCORE_EUCLIDEAN) {
            float e0 = 0.f, e1 = 0.f;
#pragma unroll 8
            for (int64_t k = 4 * i; k < d; k += 64) {
                const float4 xv = *reinterpret_cast<const float4*>(x + k), yv = *reinterpret_cast<const float4*>(y + k);
                const float t0 = xv.x - yv.x, t1 = xv.y - yv.y, t2 = xv.z - yv.z, t3 = xv.w - yv.w;
                e0 = __builtin_fmaf(t0, t0, e0); e1 = __builtin_fmaf(t1, t1, e1);
                e0 = __builtin_fmaf(t2, t2, e0); e1 = __builtin_fmaf(t3, t3, e1);
            }
            const float out = -sqrtf(rowsum(e0 + e1));
            if (i == 0 && j < N) orow[j] = out;
            continue;
        }
#pragma unroll 8
        for (int64_t k = 4 * i; k < d; k += 64) {
            const float4 xv = *reinterpret_cast<const float4*>(x + k), yv = *reinterpret_cast<const float4*>(y + k);
            a0 += xv.x * yv.x + xv.y * yv.y + xv.z * yv.z + xv.w * yv.w;
            a1 += xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
            a2 += yv.x * yv.x + yv.y * yv.y + yv.z * yv.z + yv.w * yv.w;
        }
        a0 = rowsum(a0); a1 = rowsum(a1); a2 = rowsum(a2);
        float out;
        if (KIND == NW_SCORE_HYPERSPHERE) {   // direct form on the normalised rows (the row is re-read from the caches)
            const float inq = 1.f / fmaxf(sqrtf(a1), NW_NORM_EPS), ins = 1.f / fmaxf(sqrtf(a2), NW_NORM_EPS);
            float acc = 0.f;
            for (int64_t k = 4 * i; k < d; k += 64) {
                const float4 xv = *reinterpret_cast<const float4*>(x + k), yv = *reinterpret_cast<const float4*>(y + k);
                const float t0 = xv.x * inq - yv.x * ins, t1 = xv.y * inq - yv.y * ins, t2 = xv.z * inq - yv.z * ins,
                            t3 = xv.w * inq - yv.w * ins;
                acc += t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3;
            }
            out = -sqrtf(rowsum(acc));
        } else {
            out = score_from_dot<KIND>(a0, a1, a2, scale);
        }
        if (i == 0 && j < N) orow[j] = out;
    }
}

template <int RS, int KIND>
int launch_mfma_rs(const float* q, const float* s, float* scores, const float* ls, int B, int N,
                   int d, hipStream_t st) {
    const int n_stiles = (N + 16 * RS - 1) / (16 * RS);
    const int n_qtiles = (B + BQ - 1) / BQ;
    const int grid = padded_grid(n_stiles, n_qtiles);
    const size_t lds = 1024 + TileCfg<RS>::STAGE_BYTES;
    hipLaunchKernelGGL((nw_scores_mfma_kernel<RS, KIND>), dim3(grid), dim3(TILE_THREADS), lds, st, q, s,
                       scores, ls, B, N, d, n_stiles, n_qtiles);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

template <int KIND>
int launch_kind(const float* q, const float* s, float* scores, int64_t B, int64_t N, int64_t d,
                const float* ls, int sup_batched, hipStream_t st) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(s) |
                           reinterpret_cast<uintptr_t>(scores)) & 15) == 0;
    const bool mfma_ok = !sup_batched && N > 25 && d >= 4 && (d % 4) == 0 && aligned &&
                         B < (1 << 30) && N < (1 << 30) && d < (1 << 30);
    if (mfma_ok) {
        switch (pick_rs(B, N, /*d: register-staged path, even tiles only*/ 1)) {
            case 2: return launch_mfma_rs<2, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            case 4: return launch_mfma_rs<4, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            case 6: return launch_mfma_rs<6, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            case 8: return launch_mfma_rs<8, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            case 10: return launch_mfma_rs<10, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            default: return launch_mfma_rs<12, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
        }
    }
    const int64_t pairs = B * N;
    if (sup_batched && N > 25 && d % 4 == 0 && aligned && B <= 65535) {   // per-query supports at large N: one streaming pass
        const int64_t gx = (N + 63) / 64;
        if (gx > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
        hipLaunchKernelGGL((nw_scores_batched_stream_kernel<KIND>), dim3((unsigned)gx, (unsigned)B), dim3(256), 0, st, q, s, scores,
                           ls, N, d);
        NW_CHECK_LAUNCH();
        return NW_OK;
    }
    if (B >= 4096 && N > 1) {   // enough queries to fill the chip with a wave each
        const int64_t grid = (B + 3) / 4;
        if (grid > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
        hipLaunchKernelGGL((nw_scores_direct_rows_kernel<KIND>), dim3((unsigned)grid), dim3(256), 0, st, q,
                           s, scores, ls, B, N, d, sup_batched);
        NW_CHECK_LAUNCH();
        return NW_OK;
    }
    const int64_t grid = (pairs + 3) / 4;
    if (grid > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    hipLaunchKernelGGL((nw_scores_direct_kernel<KIND>), dim3((unsigned)grid), dim3(256), 0, st, q,
                       s, scores, ls, B, N, d, sup_batched);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

}  // namespace

int launch_scores(const float* q, const float* s, float* scores, int64_t B, int64_t N, int64_t d,
                  int kind, const float* ls, int sup_batched, hipStream_t st) {
    if (B <= 0 || N <= 0) return NW_OK;
    switch (kind) {
        case NW_SCORE_EUCLIDEAN: return launch_kind<NW_SCORE_EUCLIDEAN>(q, s, scores, B, N, d, ls, sup_batched, st);
        case NW_SCORE_HYPERSPHERE: return launch_kind<NW_SCORE_HYPERSPHERE>(q, s, scores, B, N, d, ls, sup_batched, st);
        case NW_SCORE_COSINE: return launch_kind<NW_SCORE_COSINE>(q, s, scores, B, N, d, ls, sup_batched, st);
        case NW_SCORE_DOT: return launch_kind<NW_SCORE_DOT>(q, s, scores, B, N, d, ls, sup_batched, st);
        case NW_SCORE_CLIP: return launch_kind<NW_SCORE_CLIP>(q, s, scores, B, N, d, ls, sup_batched, st);
        default: return NW_ERR_UNSUPPORTED;
    }
}

}  // namespace nw
