// Internal declarations shared by the .hip translation units of libnwhead_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nwhead_hip.h"

#define NW_LOG_EPS 1e-12f   /* nwhead/nw.py:289 */
#define NW_NORM_EPS 1e-12f  /* F.normalize default eps, nwhead/kernel.py:19-20 */

#define NW_CHECK_LAUNCH()                          \
    do {                                           \
        if (hipGetLastError() != hipSuccess) return NW_ERR_LAUNCH; \
    } while (0)

namespace nw {

// scores (B,N) <- q (B,d), s (N,d) | (B,N,d)
int launch_scores(const float* q, const float* s, float* scores, int64_t B, int64_t N, int64_t d,
                  int kind, const float* logit_scale_dev, int sup_batched, hipStream_t st);

// softmax over supports + per-class aggregation of one (B,N) score matrix
//   final:   out (B,C), optional lse (B,), optional weights (B,N)
//   partial: m (B,), den (B,), num (B,C)
int launch_aggregate(const float* scores, const int64_t* sy, int labels_batched, float* out,
                     float* lse, float* weights, float* m, float* den, float* num, int64_t B,
                     int64_t N, int64_t C, hipStream_t st);

int launch_merge(const float* m, const float* den, const float* num, float* out, int64_t G,
                 int64_t B, int64_t C, int64_t sm, int64_t sd, int64_t sn, hipStream_t st);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Block-wide reductions for 256-thread workgroups (4 waves); `red` is >= 8 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    const int nw_ = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw_; ++w) r += red[w];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = -INFINITY;
    const int nw_ = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw_; ++w) r = fmaxf(r, red[w]);
    return r;
}

// score from (dot, |q|^2, |s|^2); shared by the MFMA and the generic kernels
template <int KIND>
__device__ __forceinline__ float score_from_dot(float dot, float qn2, float sn2, float scale) {
    if (KIND == NW_SCORE_DOT) return dot;
    if (KIND == NW_SCORE_EUCLIDEAN) return -sqrtf(fmaxf(qn2 + sn2 - 2.f * dot, 0.f));
    const float nq = fmaxf(sqrtf(qn2), NW_NORM_EPS), ns = fmaxf(sqrtf(sn2), NW_NORM_EPS);
    const float c = dot / (nq * ns);
    if (KIND == NW_SCORE_COSINE) return c;
    if (KIND == NW_SCORE_CLIP) return scale * c;
    // hypersphere: -||x/|x| - y/|y|||, matmul form like torch's cdist for N > 25
    const float a = qn2 / (nq * nq), b = sn2 / (ns * ns);
    return -sqrtf(fmaxf(a + b - 2.f * c, 0.f));
}

}  // namespace nw
