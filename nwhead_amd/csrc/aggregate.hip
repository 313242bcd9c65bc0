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
    float* __restrict__ m_out, float* __restrict__ den_out, float* __restrict__ num_out, int64_t Ntot,
    int64_t C, int use_ranges, int S, int64_t Ns) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // 8 floats
    float* ev = red + 8;                          // AGG_CH: exp(s - m) of the staged chunk
    int* yv = reinterpret_cast<int*>(ev + AGG_CH);  // AGG_CH: its labels (-1: outside [0, C))
    float* num = reinterpret_cast<float*>(yv + AGG_CH);  // C floats
    int* jlo = reinterpret_cast<int*>(num + C);           // C ints (use_ranges)
    int* jhi = jlo + C;                                   // C ints (use_ranges)
    // S > 1 (PARTIAL only): workgroup b S + sl takes the supports [sl Ns, (sl + 1) Ns) of query b and writes its partials at
    // index sl B + b (the layout nw_merge_kernel reads as S "shards"): a handful of queries with thousands of supports
    // each (per-query supports at large N) would otherwise run on a handful of CUs
    const int64_t b = S > 1 ? blockIdx.x / S : blockIdx.x;
    const int64_t sl = S > 1 ? blockIdx.x - b * S : 0;
    const int64_t N = S > 1 ? max((int64_t)0, min(Ns, Ntot - sl * Ns)) : Ntot;
    const int64_t pidx = S > 1 ? sl * (gridDim.x / S) + b : b;
    const int tid = threadIdx.x;
    const float* row = scores + b * Ntot + sl * Ns;
    const int64_t* lab = sy + (labels_batched ? b * Ntot : 0) + sl * Ns;

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
            if (!PARTIAL && weights) weights[b * Ntot + j] = e * inv_den;
        }
        __syncthreads();
        for (int64_t c = tid; c < C; c += 256) {
            const int k0 = use_ranges ? jlo[c] : 0, k1 = use_ranges ? jhi[c] : len - 1;
            // the class's members of the chunk, in a fixed order: four interleaved partial sums over 16-byte LDS reads
            // (labels and weights four at a time; a non-member adds +0) -- one element at a time this scan was a chain of
            // dependent LDS round trips, 0.4 ms for 16 x 5000 supports with unsorted labels
            float a = num[c], p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            int k = k0;
            for (; k <= k1 && (k & 3); ++k) a += (yv[k] == (int)c) ? ev[k] : 0.f;
#pragma unroll 4
            for (; k <= k1 - 3; k += 4) {   // (k1 - 3, not k + 3: an absent class has k0 = INT_MAX)
                const int4 y4 = *reinterpret_cast<const int4*>(yv + k);
                const float4 e4 = *reinterpret_cast<const float4*>(ev + k);
                p0 += (y4.x == (int)c) ? e4.x : 0.f;
                p1 += (y4.y == (int)c) ? e4.y : 0.f;
                p2 += (y4.z == (int)c) ? e4.z : 0.f;
                p3 += (y4.w == (int)c) ? e4.w : 0.f;
            }
            a += (p0 + p1) + (p2 + p3);
            for (; k <= k1; ++k) a += (yv[k] == (int)c) ? ev[k] : 0.f;
            num[c] = a;
        }
        __syncthreads();
    }

    if (PARTIAL) {
        if (tid == 0) {
            m_out[pidx] = m;
            den_out[pidx] = den;
        }
        for (int64_t c = tid; c < C; c += 256) num_out[pidx * C + c] = num[c];
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

int launch_merge(const float* m, const float* den, const float* num, float* out, int64_t G,
                 int64_t B, int64_t C, int64_t sm, int64_t sd, int64_t sn, const int64_t* class_lo,
                 int64_t CL, hipStream_t st);

int aggregate_slices(int64_t B, int64_t N) {   // slices per query of the sliced aggregation (1: one workgroup per query)
    if (B <= 0 || B > 64 || N < 4096) return 1;
    const int64_t by_cus = 256 / B, by_len = (N + 2047) / 2048;
    const int64_t s = by_cus < by_len ? by_cus : by_len;
    return s < 2 ? 1 : (int)s;
}

int launch_aggregate(const float* scores, const int64_t* sy, int labels_batched, float* out,
                     float* lse, float* weights, float* m, float* den, float* num, int64_t B,
                     int64_t N, int64_t C, hipStream_t st, float* slice_ws, size_t slice_ws_floats) {
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
    const int S = (!partial && !lse && !weights && slice_ws) ? aggregate_slices(B, N) : 1;
    if (S > 1 && slice_ws_floats >= (size_t)S * B * (2 + C)) {
        // few queries, long rows: S workgroups per query leave (m, den, num) partials, merged like the shards of a bank
        float* pm = slice_ws;
        float* pd = pm + (size_t)S * B;
        float* pn = pd + (size_t)S * B;
        const int64_t Ns = ((N + S - 1) / S + 3) / 4 * 4;   // (rows stay 16-byte aligned)
        hipLaunchKernelGGL(nw_aggregate_kernel<true>, dim3((unsigned)(B * S)), dim3(256), lds, st, scores, sy, labels_batched,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, pm, pd, pn, N, C, use_ranges, S, Ns);
        NW_CHECK_LAUNCH();
        return launch_merge(pm, pd, pn, out, S, B, C, B, B, B * C, nullptr, 0, st);
    }
    if (partial)
        hipLaunchKernelGGL(nw_aggregate_kernel<true>, dim3((unsigned)B), dim3(256), lds, st, scores, sy,
                           labels_batched, out, lse, weights, m, den, num, N, C, use_ranges, 1, N);
    else
        hipLaunchKernelGGL(nw_aggregate_kernel<false>, dim3((unsigned)B), dim3(256), lds, st, scores, sy,
                           labels_batched, out, lse, weights, m, den, num, N, C, use_ranges, 1, N);
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
