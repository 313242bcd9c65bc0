// backward.hip -- gradient of NWHead.forward w.r.t. queries, supports and the CLIP log-scale
// (gfx950 / MI355X only).
//
// The reference gets this from autograd through nwhead/nw.py:276-289 and nwhead/kernel.py:13-44
// (loss.backward(), train.py:414).  Closed form, written for a score
// s = f(dot, |q|^2, |s|^2):
//     dP   = g * exp(-out)                      (= g / (P + 1e-12), nw.py:289)
//     dW_j = dP[sy_j];  W_j = exp(s_j - lse);   dS_j = W_j (dW_j - sum_j W_j dW_j)
//     A_bj = dS * df/ddot;  rq_b = sum_j dS * df/d|q|^2;  rs_j = sum_b dS * df/d|s|^2
//     gq = A s + 2 rq q            gs = A^T q + 2 rs s
// Euclidean: df/ddot = 1/D, df/d|q|^2 = df/d|s|^2 = -1/(2D), all taken as 0 where D == 0 (torch's
// cdist backward masks the zero distance the same way).
#include "nw_internal.h"

namespace nw {
namespace {

__global__ __launch_bounds__(256) void nw_rownorm_kernel(const float* __restrict__ x,
                                                          float* __restrict__ n2, int64_t rows,
                                                          int64_t d) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* p = x + r * d;
    float a = 0.f;
    for (int64_t k = lane; k < d; k += 64) a += p[k] * p[k];
    a = wave_sum(a);
    if (lane == 0) n2[r] = a;
}

// One workgroup per query.  Writes A (B,N), Rs (B,N), rq (B,), gls (B,).
template <int KIND>
__global__ __launch_bounds__(256) void nw_bwd_coeff_kernel(
    const float* __restrict__ scores, const float* __restrict__ lse, const float* __restrict__ out,
    const float* __restrict__ gout, const int64_t* __restrict__ sy, int labels_batched,
    const float* __restrict__ qn2, const float* __restrict__ sn2, int sup_batched,
    const float* __restrict__ logit_scale, float* __restrict__ A, float* __restrict__ Rs,
    float* __restrict__ rq, float* __restrict__ gls, int64_t N, int64_t C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    float* dP = red + 8;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const float* row = scores + b * N;
    const int64_t* lab = sy + (labels_batched ? b * N : 0);
    const float l = lse[b];

    for (int64_t c = tid; c < C; c += 256) dP[c] = gout[b * C + c] * expf(-out[b * C + c]);
    __syncthreads();

    float t = 0.f;
    for (int64_t j = tid; j < N; j += 256) {
        const int64_t y = lab[j];
        const float dw = ((uint64_t)y < (uint64_t)C) ? dP[y] : 0.f;
        t += expf(row[j] - l) * dw;
    }
    t = block_sum(t, red);

    float scale = 1.f, nq = 1.f, inq2 = 0.f;
    if (KIND == NW_SCORE_CLIP) scale = expf(*logit_scale);
    if (KIND == NW_SCORE_HYPERSPHERE || KIND == NW_SCORE_COSINE || KIND == NW_SCORE_CLIP) {
        const float n = sqrtf(qn2[b]);
        nq = fmaxf(n, NW_NORM_EPS);
        inq2 = (n > NW_NORM_EPS) ? 1.f / (nq * nq) : 0.f;  // F.normalize clamps: no grad via |q|
    }
    float rq_acc = 0.f, gls_acc = 0.f;
    for (int64_t j = tid; j < N; j += 256) {
        const float sc = row[j];
        const int64_t y = lab[j];
        const float dw = ((uint64_t)y < (uint64_t)C) ? dP[y] : 0.f;
        const float dS = expf(sc - l) * (dw - t);
        float a, r;
        if (KIND == NW_SCORE_DOT) {
            a = dS;
            r = 0.f;
        } else if (KIND == NW_SCORE_EUCLIDEAN) {
            const float D = -sc;
            a = (D == 0.f) ? 0.f : dS / D;
            r = -0.5f * a;
            rq_acc += r;
        } else {
            const float sn = sqrtf(sn2[sup_batched ? b * N + j : j]);
            const float ns = fmaxf(sn, NW_NORM_EPS);
            const float ins2 = (sn > NW_NORM_EPS) ? 1.f / (ns * ns) : 0.f;
            float fc, c;  // df/dcos and cos
            if (KIND == NW_SCORE_HYPERSPHERE) {
                const float D = -sc;
                fc = (D == 0.f) ? 0.f : 1.f / D;
                c = 1.f - 0.5f * D * D;
            } else {
                fc = scale;
                c = sc / scale;
                gls_acc += dS * sc;  // d(e^ls cos)/d ls = score
            }
            const float gc = dS * fc;
            a = gc / (nq * ns);
            rq_acc += gc * (-0.5f * c * inq2);
            r = gc * (-0.5f * c * ins2);
        }
        A[b * N + j] = a;
        Rs[b * N + j] = r;
    }
    rq_acc = block_sum(rq_acc, red);
    gls_acc = block_sum(gls_acc, red);
    if (tid == 0) {
        rq[b] = rq_acc;
        gls[b] = gls_acc;
    }
}

// gq[b,k] = sum_j A[b,j] s[(b,)j,k] + 2 rq[b] q[b,k]
__global__ __launch_bounds__(256) void nw_bwd_gq_kernel(const float* __restrict__ A,
                                                         const float* __restrict__ rq,
                                                         const float* __restrict__ q,
                                                         const float* __restrict__ s,
                                                         float* __restrict__ gq, int64_t N,
                                                         int64_t d, int sup_batched) {
    const int64_t b = blockIdx.x;
    const int64_t k = (int64_t)blockIdx.y * 256 + threadIdx.x;
    if (k >= d) return;
    const float* a = A + b * N;
    const float* sb = s + (sup_batched ? b * N * d : 0);
    float acc = 0.f;
    for (int64_t j = 0; j < N; ++j) acc += a[j] * sb[j * d + k];
    gq[b * d + k] = acc + 2.f * rq[b] * q[b * d + k];
}

// shared support: gs[j,k] = sum_b A[b,j] q[b,k] + 2 (sum_b Rs[b,j]) s[j,k]
__global__ __launch_bounds__(256) void nw_bwd_gs_shared_kernel(const float* __restrict__ A,
                                                                const float* __restrict__ Rs,
                                                                const float* __restrict__ q,
                                                                const float* __restrict__ s,
                                                                float* __restrict__ gs, int64_t B,
                                                                int64_t N, int64_t d) {
    const int64_t j = blockIdx.x;
    const int64_t k = (int64_t)blockIdx.y * 256 + threadIdx.x;
    if (k >= d) return;
    float acc = 0.f, rs = 0.f;
    for (int64_t b = 0; b < B; ++b) {
        acc += A[b * N + j] * q[b * d + k];
        rs += Rs[b * N + j];
    }
    gs[j * d + k] = acc + 2.f * rs * s[j * d + k];
}

// per-query support: gs[b,j,k] = A[b,j] q[b,k] + 2 Rs[b,j] s[b,j,k]
__global__ __launch_bounds__(256) void nw_bwd_gs_batched_kernel(const float* __restrict__ A,
                                                                 const float* __restrict__ Rs,
                                                                 const float* __restrict__ q,
                                                                 const float* __restrict__ s,
                                                                 float* __restrict__ gs, int64_t N,
                                                                 int64_t d) {
    const int64_t bj = blockIdx.x;
    const int64_t b = bj / N;
    const int64_t k = (int64_t)blockIdx.y * 256 + threadIdx.x;
    if (k >= d) return;
    gs[bj * d + k] = A[bj] * q[b * d + k] + 2.f * Rs[bj] * s[bj * d + k];
}

__global__ __launch_bounds__(256) void nw_sum_kernel(const float* __restrict__ x,
                                                      float* __restrict__ out, int64_t n) {
    __shared__ float red[8];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) a += x[i];
    a = block_sum(a, red);
    if (threadIdx.x == 0) *out = a;
}

struct BwdWs {
    float *A, *Rs, *rq, *gls, *qn2, *sn2;
};
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t bwd_layout(int64_t B, int64_t N, int sup_batched, char* base, BwdWs* ws) {
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += align256(nfloat * sizeof(float));
        return p;
    };
    BwdWs w;
    w.A = take((size_t)B * N);
    w.Rs = take((size_t)B * N);
    w.rq = take((size_t)B);
    w.gls = take((size_t)B);
    w.qn2 = take((size_t)B);
    w.sn2 = take(sup_batched ? (size_t)B * N : (size_t)N);
    if (ws) *ws = w;
    return off;
}

}  // namespace

int launch_rownorm2(const float* x, float* n2, int64_t rows, int64_t d, hipStream_t st) {
    if (rows <= 0) return NW_OK;
    if ((rows + 3) / 4 > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    hipLaunchKernelGGL(nw_rownorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, n2, rows, d);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

}  // namespace nw

extern "C" size_t nw_bwd_workspace_bytes(int64_t B, int64_t N, int64_t d, int64_t C, int kind,
                                         int sup_batched) {
    (void)d; (void)C; (void)kind;
    if (B <= 0 || N < 0) return 0;
    return nw::bwd_layout(B, N, sup_batched, nullptr, nullptr);
}

extern "C" int nw_bwd_f32(const float* q, const float* s, const int64_t* sy, const float* scores,
                          const float* lse, const float* out, const float* gout, float* gq,
                          float* gs, float* glogit_scale, void* workspace, size_t workspace_bytes,
                          int64_t B, int64_t N, int64_t d, int64_t C, int kind,
                          const float* logit_scale_dev, int sup_batched, int labels_batched,
                          void* stream) {
    using namespace nw;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B < 0 || N < 0 || d < 0 || C < 0) return NW_ERR_INVALID_ARG;
    if (kind < NW_SCORE_EUCLIDEAN || kind > NW_SCORE_CLIP) return NW_ERR_UNSUPPORTED;
    if (kind == NW_SCORE_CLIP && !logit_scale_dev) return NW_ERR_INVALID_ARG;
    const int64_t gs_rows = sup_batched ? B * N : N;
    if (B == 0 || N == 0 || d == 0) {
        if (gq && B * d) if (hipMemsetAsync(gq, 0, (size_t)B * d * 4, st) != hipSuccess) return NW_ERR_LAUNCH;
        if (gs && gs_rows * d) if (hipMemsetAsync(gs, 0, (size_t)gs_rows * d * 4, st) != hipSuccess) return NW_ERR_LAUNCH;
        if (glogit_scale) if (hipMemsetAsync(glogit_scale, 0, 4, st) != hipSuccess) return NW_ERR_LAUNCH;
        return NW_OK;
    }
    if (!q || !s || !sy || !scores || !lse || !out || !gout || !gq || !gs) return NW_ERR_INVALID_ARG;
    if (B > 0x7fffffffLL || gs_rows > 0x7fffffffLL || (d + 255) / 256 > 65535) return NW_ERR_INVALID_ARG;
    BwdWs ws;
    const size_t need = bwd_layout(B, N, sup_batched, static_cast<char*>(workspace), &ws);
    if (!workspace || workspace_bytes < need) return NW_ERR_WORKSPACE;
    const size_t lds = (8 + (size_t)C) * sizeof(float);
    if (lds > 160 * 1024) return NW_ERR_UNSUPPORTED;

    const bool norms = (kind == NW_SCORE_HYPERSPHERE || kind == NW_SCORE_COSINE || kind == NW_SCORE_CLIP);
    if (norms) {
        hipLaunchKernelGGL(nw_rownorm_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, q, ws.qn2, B, d);
        hipLaunchKernelGGL(nw_rownorm_kernel, dim3((unsigned)((gs_rows + 3) / 4)), dim3(256), 0, st, s, ws.sn2, gs_rows, d);
    }
#define NW_COEFF(K)                                                                              \
    hipLaunchKernelGGL(nw_bwd_coeff_kernel<K>, dim3((unsigned)B), dim3(256), lds, st, scores, lse, \
                       out, gout, sy, labels_batched, ws.qn2, ws.sn2, sup_batched, logit_scale_dev, \
                       ws.A, ws.Rs, ws.rq, ws.gls, N, C)
    switch (kind) {
        case NW_SCORE_EUCLIDEAN: NW_COEFF(NW_SCORE_EUCLIDEAN); break;
        case NW_SCORE_HYPERSPHERE: NW_COEFF(NW_SCORE_HYPERSPHERE); break;
        case NW_SCORE_COSINE: NW_COEFF(NW_SCORE_COSINE); break;
        case NW_SCORE_DOT: NW_COEFF(NW_SCORE_DOT); break;
        default: NW_COEFF(NW_SCORE_CLIP); break;
    }
#undef NW_COEFF
    const unsigned kd = (unsigned)((d + 255) / 256);
    hipLaunchKernelGGL(nw_bwd_gq_kernel, dim3((unsigned)B, kd), dim3(256), 0, st, ws.A, ws.rq, q, s, gq, N, d, sup_batched);
    if (sup_batched)
        hipLaunchKernelGGL(nw_bwd_gs_batched_kernel, dim3((unsigned)gs_rows, kd), dim3(256), 0, st, ws.A, ws.Rs, q, s, gs, N, d);
    else
        hipLaunchKernelGGL(nw_bwd_gs_shared_kernel, dim3((unsigned)N, kd), dim3(256), 0, st, ws.A, ws.Rs, q, s, gs, B, N, d);
    if (glogit_scale) {
        if (kind == NW_SCORE_CLIP)
            hipLaunchKernelGGL(nw_sum_kernel, dim3(1), dim3(256), 0, st, ws.gls, glogit_scale, B);
        else if (hipMemsetAsync(glogit_scale, 0, 4, st) != hipSuccess)
            return NW_ERR_LAUNCH;
    }
    NW_CHECK_LAUNCH();
    return NW_OK;
}
