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
#include "nw_internal.h"

namespace nw {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BQ = 64;   // queries per workgroup: 4 waves x one 16-wide MFMA column block
constexpr int BK = 32;   // k per pipeline stage = one 128-byte line per row
constexpr int ROW_F4 = BK / 4;

// LDS rows are 128 B = eight 16-byte slots.  ds_read_b128 is serviced in 16-lane groups over a
// 256-B bank row, so rows r and r+2 collide slot for slot; XOR the slot with (row >> 1) & 7.
__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// Workgroup tile = 64 queries x (16*RS) supports, K streamed in 32-float stages through a
// double-buffered LDS image (register-staged: global_load_dwordx4 -> ds_write_b128).
// Wave w owns query columns [16w, 16w+16) and all RS support blocks of the tile:
//   acc[rs][r] = dot(support 16*rs + 4*(lane>>4) + r, query 16*w + (lane&15)).
// MFMA operand map (16x16x4 f32): lane l supplies A[row l&15][k l>>4], B[k l>>4][col l&15]; the
// k order is free, so lane group g takes the four consecutive floats k = 16t + 4g .. +3 of its row
// (one ds_read_b128) and feeds them to four consecutive MFMAs.
template <int RS, int KIND>
__global__ __launch_bounds__(256) void nw_scores_mfma_kernel(
    const float* __restrict__ q, const float* __restrict__ s, float* __restrict__ scores,
    const float* __restrict__ logit_scale, int B, int N, int d, int n_stiles, int n_qtiles) {
    constexpr int BS = 16 * RS;
    constexpr int TILE_F4 = (BQ + BS) * ROW_F4;
    constexpr bool NEED_NORM = (KIND != NW_SCORE_DOT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* lds = reinterpret_cast<float4*>(smem);

    // XCD-aware decode: the n_qtiles workgroups that stream the same support tile get block ids
    // that are equal mod 8, i.e. the same XCD / L2 under round-robin dispatch (speed only).
    const int per_grp = 8 * n_qtiles;
    const int grp = blockIdx.x / per_grp, rem = blockIdx.x % per_grp;
    const int qt = rem >> 3, st = grp * 8 + (rem & 7);
    if (st >= n_stiles) return;
    const int q0 = qt * BQ, s0 = st * BS;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, g = lane >> 4;

    // ---- staging assignment: 16-byte chunk c -> (row c>>3, slot c&7)
    float4 rq[2], rsg[RS / 2];
    auto gload = [&](int kt) {
        const int kb = kt * BK;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = tid + 256 * it, row = c >> 3, k = kb + (c & 7) * 4;
            const int gr = min(q0 + row, B - 1);
            rq[it] = (k < d) ? *reinterpret_cast<const float4*>(q + (size_t)gr * d + k)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < RS / 2; ++it) {
            const int c = tid + 256 * it, row = c >> 3, k = kb + (c & 7) * 4;
            const int gr = min(s0 + row, N - 1);
            rsg[it] = (k < d) ? *reinterpret_cast<const float4*>(s + (size_t)gr * d + k)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
        float4* Qs = lds + buf * TILE_F4;
        float4* Ss = Qs + BQ * ROW_F4;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = tid + 256 * it, row = c >> 3;
            Qs[row * ROW_F4 + swz(row, c & 7)] = rq[it];
        }
#pragma unroll
        for (int it = 0; it < RS / 2; ++it) {
            const int c = tid + 256 * it, row = c >> 3;
            Ss[row * ROW_F4 + swz(row, c & 7)] = rsg[it];
        }
    };

    f32x4 acc[RS];
    float sqs[RS];
    float sqq = 0.f;
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        sqs[r] = 0.f;
    }

    const int nk = (d + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();

    const int qrow = 16 * wave + i;
    const int rsw = (i >> 1) & 7;  // == ((16*rs + i) >> 1) & 7 for every rs; qrow likewise
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float4* Qs = lds + buf * TILE_F4;
        const float4* Ss = Qs + BQ * ROW_F4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int slot = (4 * t + g) ^ rsw;
            const float4 b = Qs[qrow * ROW_F4 + slot];
            if (NEED_NORM) sqq += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                const float4 a = Ss[(16 * r + i) * ROW_F4 + slot];
                if (NEED_NORM) sqs[r] += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[r], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: norms -> scores
    float scale = 1.f;
    if (KIND == NW_SCORE_CLIP) scale = expf(*logit_scale);
    float* sn = reinterpret_cast<float*>(smem);  // [BS] support squared norms (staging is dead)
    if (NEED_NORM) {
        sqq += __shfl_xor(sqq, 16);
        sqq += __shfl_xor(sqq, 32);
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            float v = sqs[r];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if ((r & 3) == wave && g == 0) sn[16 * r + i] = v;  // every wave holds the same value
        }
        __syncthreads();
    }
    const int b = q0 + qrow;
    if (b >= B) return;
    float* orow = scores + (size_t)b * N;
    const bool vec_ok = (N & 3) == 0;
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int j = s0 + 16 * r + 4 * g;
        if (j >= N) continue;
        float4 n2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (NEED_NORM) n2 = *reinterpret_cast<const float4*>(sn + 16 * r + 4 * g);
        float4 o;
        o.x = score_from_dot<KIND>(acc[r][0], sqq, n2.x, scale);
        o.y = score_from_dot<KIND>(acc[r][1], sqq, n2.y, scale);
        o.z = score_from_dot<KIND>(acc[r][2], sqq, n2.z, scale);
        o.w = score_from_dot<KIND>(acc[r][3], sqq, n2.w, scale);
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
__global__ __launch_bounds__(256) void nw_scores_direct_kernel(
    const float* __restrict__ q, const float* __restrict__ s, float* __restrict__ scores,
    const float* __restrict__ logit_scale, int64_t B, int64_t N, int64_t d, int sup_batched) {
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pair >= B * N) return;
    const int64_t b = pair / N, j = pair % N;
    const float* x = q + b * d;
    const float* y = s + (sup_batched ? (b * N + j) * d : j * d);
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

// Pick the support-tile height (in 16-row blocks) that minimises the number of workgroup rounds
// on 256 CUs times the per-workgroup cost (RS blocks of MFMA work + a fixed overhead).
int pick_rs(int64_t B, int64_t N) {
    const int cand[] = {2, 4, 6, 8, 10, 12};
    const int64_t nq = (B + BQ - 1) / BQ;
    double best = 1e30;
    int best_rs = 8;
    for (int rs : cand) {
        const int64_t ns = (N + 16 * rs - 1) / (16 * rs);
        const int64_t rounds = (nq * ns + 255) / 256;
        const double cost = (double)rounds * (rs + 1.5);
        if (cost < best - 1e-9) {
            best = cost;
            best_rs = rs;
        }
    }
    return best_rs;
}

template <int RS, int KIND>
int launch_mfma_rs(const float* q, const float* s, float* scores, const float* ls, int B, int N,
                   int d, hipStream_t st) {
    const int n_stiles = (N + 16 * RS - 1) / (16 * RS);
    const int n_qtiles = (B + BQ - 1) / BQ;
    const int grid = ((n_stiles + 7) / 8) * 8 * n_qtiles;
    const size_t lds = (size_t)2 * (BQ + 16 * RS) * BK * sizeof(float);
    hipLaunchKernelGGL((nw_scores_mfma_kernel<RS, KIND>), dim3(grid), dim3(256), lds, st, q, s,
                       scores, ls, B, N, d, n_stiles, n_qtiles);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

template <int KIND>
int launch_kind(const float* q, const float* s, float* scores, int64_t B, int64_t N, int64_t d,
                const float* ls, int sup_batched, hipStream_t st) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(s) |
                           reinterpret_cast<uintptr_t>(scores)) & 15) == 0;
    const bool mfma_ok = !sup_batched && N > 25 && (d % 4) == 0 && aligned &&
                         B < (1 << 30) && N < (1 << 30) && d < (1 << 30);
    if (mfma_ok) {
        switch (pick_rs(B, N)) {
            case 2: return launch_mfma_rs<2, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            case 4: return launch_mfma_rs<4, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            case 6: return launch_mfma_rs<6, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            case 8: return launch_mfma_rs<8, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            case 10: return launch_mfma_rs<10, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
            default: return launch_mfma_rs<12, KIND>(q, s, scores, ls, (int)B, (int)N, (int)d, st);
        }
    }
    const int64_t pairs = B * N;
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
