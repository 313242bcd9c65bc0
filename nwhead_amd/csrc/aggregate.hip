// aggregate.hip -- row softmax over supports + label aggregation + log (gfx950 / MI355X only).
//
// Replaces nwhead/nw.py:276 (one_hot), :285 (softmax), :287 (bmm with the one-hot) and :289
// (log(. + 1e-12)) without ever forming the (B,N,C) one-hot: each support's weight is added to the
// accumulator of its class index.  Also emits the per-shard (m, den, num) partials of SURVEY 8e
// and merges them (nw_merge_finalize).
#include "nw_internal.h"

namespace nw {
namespace {

// One 256-thread workgroup per query row.  HBM-bound streaming of one score row (4*N bytes, re-read
// from L2 on the later passes) plus 8*N bytes of labels.
//   pass 1: m = max_j s_j          pass 2: den = sum_j exp(s_j - m)
//   pass 3: per-class sums of exp(s_j - m), optional normalised weights.  No float atomics: the row is
//           staged through LDS in chunks of AGG_CH supports, every class is owned by ONE thread, which adds
//           the chunk's members of its class in support order (it scans [first, last] position of the class
//           inside the chunk, found with integer LDS min / max) -- the result is bit-reproducible.
constexpr int AGG_CH = 2048;
template <bool PARTIAL>
__global__ __launch_bounds__(256) void nw_aggregate_kernel(
    const float* __restrict__ scores, const int64_t* __restrict__ sy, int labels_batched,
    float* __restrict__ out, float* __restrict__ lse, float* __restrict__ weights,
    float* __restrict__ m_out, float* __restrict__ den_out, float* __restrict__ num_out, int64_t N,
    int64_t C, int use_ranges) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // 8 floats
    float* ev = red + 8;                          // AGG_CH: exp(s - m) of the staged chunk
    int* yv = reinterpret_cast<int*>(ev + AGG_CH);  // AGG_CH: its labels (-1: outside [0, C))
    float* num = reinterpret_cast<float*>(yv + AGG_CH);  // C floats
    int* jlo = reinterpret_cast<int*>(num + C);           // C ints (use_ranges)
    int* jhi = jlo + C;                                   // C ints (use_ranges)
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const float* row = scores + b * N;
    const int64_t* lab = sy + (labels_batched ? b * N : 0);

    for (int64_t c = tid; c < C; c += 256) num[c] = 0.f;

    const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    float m = -INFINITY;
    if (vec) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
        for (int64_t j = tid; j < N / 4; j += 256) {
            const float4 v = r4[j];
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
    } else {
        for (int64_t j = tid; j < N; j += 256) m = fmaxf(m, row[j]);
    }
    m = block_max(m, red);

    float den = 0.f;
    if (vec) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
        for (int64_t j = tid; j < N / 4; j += 256) {
            const float4 v = r4[j];
            den += (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m));
        }
    } else {
        for (int64_t j = tid; j < N; j += 256) den += expf(row[j] - m);
    }
    den = block_sum(den, red);  // the barriers inside also publish the zeroed num[]
    const float inv_den = 1.f / den;

    for (int64_t base = 0; base < N; base += AGG_CH) {
        const int len = (int)((N - base < AGG_CH) ? (N - base) : AGG_CH);
        if (use_ranges) {
            for (int64_t c = tid; c < C; c += 256) {
                jlo[c] = 0x7fffffff;
                jhi[c] = -1;
            }
            __syncthreads();
        }
        for (int k = tid; k < len; k += 256) {
            const int64_t j = base + k;
            const float e = expf(row[j] - m);
            const int64_t y = lab[j];
            const int yi = ((uint64_t)y < (uint64_t)C) ? (int)y : -1;
            ev[k] = e;
            yv[k] = yi;
            if (use_ranges && yi >= 0) {
                atomicMin(&jlo[yi], k);
                atomicMax(&jhi[yi], k);
            }
            if (!PARTIAL && weights) weights[b * N + j] = e * inv_den;
        }
        __syncthreads();
        for (int64_t c = tid; c < C; c += 256) {
            const int k0 = use_ranges ? jlo[c] : 0, k1 = use_ranges ? jhi[c] : len - 1;
            float a = num[c];
            for (int k = k0; k <= k1; ++k)
                if (yv[k] == (int)c) a += ev[k];
            num[c] = a;
        }
        __syncthreads();
    }

    if (PARTIAL) {
        if (tid == 0) {
            m_out[b] = m;
            den_out[b] = den;
        }
        for (int64_t c = tid; c < C; c += 256) num_out[b * C + c] = num[c];
    } else {
        if (tid == 0 && lse) lse[b] = m + logf(den);
        for (int64_t c = tid; c < C; c += 256) out[b * C + c] = logf(num[c] * inv_den + NW_LOG_EPS);
    }
}

// N == 0: softmax over nothing, bmm gives zeros, log(0 + 1e-12)
__global__ void nw_fill_kernel(float* __restrict__ p, float v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// out[b,c] = log( sum_g num[g,b,c] e^(m_g - M) / sum_g den[g,b] e^(m_g - M) + 1e-12 )
// With class_lo != nullptr shard g only carries the CL classes [class_lo[g], class_lo[g] + CL) (a slice
// of a class-sorted bank holds ~C/G classes: the exchanged rows shrink G-fold) and num is (B, CL).
__global__ __launch_bounds__(256) void nw_merge_kernel(const float* __restrict__ m,
                                                        const float* __restrict__ den,
                                                        const float* __restrict__ num,
                                                        float* __restrict__ out, int64_t G,
                                                        int64_t B, int64_t C, int64_t sm,
                                                        int64_t sd, int64_t sn,
                                                        const int64_t* __restrict__ class_lo,
                                                        int64_t CL) {
    const int64_t b = blockIdx.x;
    float M = -INFINITY;
    for (int64_t g = 0; g < G; ++g) M = fmaxf(M, m[g * sm + b]);
    float D = 0.f;
    for (int64_t g = 0; g < G; ++g) {
        const float mg = m[g * sm + b];
        if (mg > -INFINITY) D += den[g * sd + b] * expf(mg - M);
    }
    const float inv = (D > 0.f) ? 1.f / D : 0.f;  // every shard empty: log(0 + 1e-12) like N == 0
    for (int64_t c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f;
        for (int64_t g = 0; g < G; ++g) {
            const float mg = m[g * sm + b];
            if (!(mg > -INFINITY)) continue;
            if (class_lo) {
                const int64_t j = c - class_lo[g];
                if (j >= 0 && j < CL) a += num[g * sn + b * CL + j] * expf(mg - M);
            } else {
                a += num[g * sn + b * C + c] * expf(mg - M);
            }
        }
        out[b * C + c] = logf(a * inv + NW_LOG_EPS);
    }
}

}  // namespace

int launch_aggregate(const float* scores, const int64_t* sy, int labels_batched, float* out,
                     float* lse, float* weights, float* m, float* den, float* num, int64_t B,
                     int64_t N, int64_t C, hipStream_t st) {
    if (B <= 0) return NW_OK;
    if (B > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    const bool partial = (out == nullptr);
    const size_t lds_min = (8 + 2 * (size_t)AGG_CH + (size_t)C) * sizeof(float);
    if (lds_min > 160 * 1024) return NW_ERR_UNSUPPORTED;
    const int use_ranges = lds_min + 2 * (size_t)C * sizeof(int) <= 160 * 1024;  // per-class [first, last] position tables
    const size_t lds = lds_min + (use_ranges ? 2 * (size_t)C * sizeof(int) : 0);
    if (N == 0) {
        if (partial) {
            const int64_t n = B * C;
            hipLaunchKernelGGL(nw_fill_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, m, -INFINITY, B);
            hipLaunchKernelGGL(nw_fill_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, den, 0.f, B);
            if (n) hipLaunchKernelGGL(nw_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, num, 0.f, n);
        } else {
            const int64_t n = B * C;
            if (n) hipLaunchKernelGGL(nw_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, logf(NW_LOG_EPS), n);
            if (lse) hipLaunchKernelGGL(nw_fill_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, lse, -INFINITY, B);
        }
        NW_CHECK_LAUNCH();
        return NW_OK;
    }
    if (partial)
        hipLaunchKernelGGL(nw_aggregate_kernel<true>, dim3((unsigned)B), dim3(256), lds, st, scores, sy,
                           labels_batched, out, lse, weights, m, den, num, N, C, use_ranges);
    else
        hipLaunchKernelGGL(nw_aggregate_kernel<false>, dim3((unsigned)B), dim3(256), lds, st, scores, sy,
                           labels_batched, out, lse, weights, m, den, num, N, C, use_ranges);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

int launch_merge(const float* m, const float* den, const float* num, float* out, int64_t G,
                 int64_t B, int64_t C, int64_t sm, int64_t sd, int64_t sn, const int64_t* class_lo,
                 int64_t CL, hipStream_t st) {
    if (B <= 0 || C <= 0) return NW_OK;
    if (B > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    hipLaunchKernelGGL(nw_merge_kernel, dim3((unsigned)B), dim3(256), 0, st, m, den, num, out, G, B, C, sm, sd, sn, class_lo, CL);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

}  // namespace nw
