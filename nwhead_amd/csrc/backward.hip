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
#include "tile_dma.h"
#include <cstdlib>

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

// One workgroup (256 or 1024 threads) per query.  Writes A (B,N), Rs (B,N), rq (B,), gls (B,).
// (t = sum_j W_j dW_j could be had without a pass over the supports, as sum_c g[c] (1 - 1e-12 exp(-out[c]));
//  but for a class the support set lacks that is 1 - 1 computed through a rounded `out`, ~1e-6 g instead
//  of the exact 0 the sum gives -- measured 26 us saved at B=256, N=10000, not taken.)
// SPLIT: A leaves as the split-row image A' of bwd_split.hip (row b scaled by 2^(E_b - e_j), s_scale[j] = 2^-e_j),
// with ascale[b] = 2^-E_b and qv[b] = max_k |q[b,k]| 2^-E_b; the row is staged in LDS (ld floats) for that.
template <int KIND, bool SPLIT = false>
__global__ __launch_bounds__(1024) void nw_bwd_coeff_kernel(
    const float* __restrict__ scores, const float* __restrict__ lse, const float* __restrict__ out,
    const float* __restrict__ gout, const int64_t* __restrict__ sy, int labels_batched,
    const float* __restrict__ qn2, const float* __restrict__ sn2, int sup_batched,
    const float* __restrict__ logit_scale, float* __restrict__ A, float* __restrict__ Rs,
    float* __restrict__ rq, float* __restrict__ gls, int64_t N, int64_t C, int64_t ld,
    const float* __restrict__ q = nullptr, int64_t d = 0, const float* __restrict__ s_scale = nullptr,
    float* __restrict__ ascale = nullptr, float* __restrict__ qv = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);    // 16 floats for block_sum, 64 (redx) for the combined reduction
    float* redx = red + 16;
    float* rowbuf = red + 80;                       // SPLIT: ld floats (16-byte aligned), then dP
    float* dP = SPLIT ? rowbuf + ld : red + 80;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const float* row = scores + b * N;
    const int64_t* lab = sy + (labels_batched ? b * N : 0);
    const float l = lse[b];
    float qm = 0.f;   // SPLIT: this thread's share of max_k |q[b,k]| (loads issued before anything waits)
    if (SPLIT)
        for (int64_t k = tid; k < d; k += nthr) qm = fmaxf(qm, fabsf(q[b * d + k]));

    for (int64_t c = tid; c < C; c += nthr) dP[c] = gout[b * C + c] * expf(-out[b * C + c]);
    __syncthreads();

    // Both passes over the row take four elements per thread at a time, all loads first: a plain strided loop exposes
    // one memory round trip per element (ten per thread at N = 10000), which was most of this kernel's time.
    constexpr int U = 4;
    float t = 0.f;
    for (int64_t j0 = tid; j0 < N; j0 += (int64_t)U * nthr) {
        float sc[U];
        int64_t y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = min(j0 + (int64_t)u * nthr, N - 1);
            sc[u] = row[j];
            y[u] = lab[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = j0 + (int64_t)u * nthr;
            if (j >= N) continue;
            const float dw = ((uint64_t)y[u] < (uint64_t)C) ? dP[y[u]] : 0.f;
            t += expf(sc[u] - l) * dw;
            if (SPLIT) rowbuf[j] = sc[u];   // the second pass reads the scores from LDS (same thread, same j)
        }
    }
    t = block_sum(t, red);

    float scale = 1.f, nq = 1.f, inq2 = 0.f;
    if (KIND == NW_SCORE_CLIP) scale = expf(*logit_scale);
    if (KIND == NW_SCORE_HYPERSPHERE || KIND == NW_SCORE_COSINE || KIND == NW_SCORE_CLIP) {
        const float n = sqrtf(qn2[b]);
        nq = fmaxf(n, NW_NORM_EPS);
        inq2 = (n > NW_NORM_EPS) ? 1.f / (nq * nq) : 0.f;  // F.normalize clamps: no grad via |q|
    }
    constexpr bool NORMS = (KIND == NW_SCORE_HYPERSPHERE || KIND == NW_SCORE_COSINE || KIND == NW_SCORE_CLIP);
    float rq_acc = 0.f, gls_acc = 0.f, amax = 0.f;
    for (int64_t j0 = tid; j0 < N; j0 += (int64_t)U * nthr) {
        float scv[U], snv[U], ssc[U];
        int64_t y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = min(j0 + (int64_t)u * nthr, N - 1);
            scv[u] = SPLIT ? rowbuf[j] : row[j];
            y[u] = lab[j];
            snv[u] = NORMS ? sn2[sup_batched ? b * N + j : j] : 0.f;
            ssc[u] = SPLIT ? s_scale[j] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = j0 + (int64_t)u * nthr;
            if (j >= N) continue;
            const float sc = scv[u];
            const float dw = ((uint64_t)y[u] < (uint64_t)C) ? dP[y[u]] : 0.f;
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
                const float sn = sqrtf(snv[u]);
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
            if (SPLIT) {
                const float ah = a * ssc[u];
                rowbuf[j] = ah;
                amax = fmaxf(amax, fabsf(ah));
            } else {
                A[b * ld + j] = a;
            }
            if (Rs) Rs[b * ld + j] = r;
        }
    }
    if (SPLIT)
        for (int64_t j = N + tid; j < ld; j += nthr) rowbuf[j] = 0.f;   // the K padding of the first product
    // the four block-wide results in ONE round (one barrier instead of eight; it also publishes rowbuf): per-wave
    // values into the 4 x 16 table `redx`
    rq_acc = wave_sum(rq_acc);
    gls_acc = wave_sum(gls_acc);
    amax = wave_max(amax);
    qm = wave_max(qm);
    {
        const int w = tid >> 6;
        if ((tid & 63) == 0) {
            redx[w] = rq_acc;
            redx[16 + w] = gls_acc;
            redx[32 + w] = amax;
            redx[48 + w] = qm;
        }
    }
    __syncthreads();
    {
        const int nwv = (nthr + 63) >> 6;
        rq_acc = gls_acc = 0.f;
        amax = qm = 0.f;
        for (int w = 0; w < nwv; ++w) {
            rq_acc += redx[w];
            gls_acc += redx[16 + w];
            amax = fmaxf(amax, redx[32 + w]);
            qm = fmaxf(qm, redx[48 + w]);
        }
    }
    if (tid == 0 && rq) {
        rq[b] = rq_acc;
        gls[b] = gls_acc;
    }
    if (SPLIT) {
        const int E = split_exponent(amax);
        // A row of zeros (a query whose target class no support carries, or a saturated softmax: dS == 0 exactly) must not
        // take part in the choice of G: with E = 0 its max|q| 2^-E would be the batch's largest by thirty binary orders and
        // every other row of q'' would underflow.  Its scale is 0: q'' row zero, and 0 * (A' s') = 0 in the first product.
        const float up = __builtin_ldexpf(1.f, E), down = amax > 0.f ? __builtin_ldexpf(1.f, -E) : 0.f;
        if (tid == 0) {
            ascale[b] = down;
            qv[b] = qm * down;
        }
        _Float16* dst = reinterpret_cast<_Float16*>(A + b * ld);
        for (int64_t u = tid; u < ld / 8; u += nthr) {   // eight coefficients -> 16 bytes of h + 16 bytes of l
            const float4 v0 = *reinterpret_cast<const float4*>(rowbuf + 8 * u);
            const float4 v1 = *reinterpret_cast<const float4*>(rowbuf + 8 * u + 4);
            const float x[8] = {v0.x * up, v0.y * up, v0.z * up, v0.w * up, v1.x * up, v1.y * up, v1.z * up, v1.w * up};
            half8 h, l;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                h[k] = (_Float16)x[k];
                l[k] = (_Float16)(x[k] - (float)h[k]);
            }
            const int64_t chunk = (8 * u) >> 5, within = (8 * u) & 31;
            *reinterpret_cast<half8*>(dst + chunk * 64 + within) = h;
            *reinterpret_cast<half8*>(dst + chunk * 64 + 32 + within) = l;
        }
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

// ---------------------------------------------------------------------------------------------------
// The two products of the backward on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMAs):
//     gq[b,k] = sum_j A[b,j] s[j,k] + 2 rq[b] q[b,k]        (M = B, K = N; A-operand is K-contiguous)
//     gs[j,k] = sum_b A[b,j] q[b,k] + 2 rs[j] s[j,k]        (M = N, K = B; A-operand is M-contiguous)
// One kernel: C[m,n] = sum_k Aop(m,k) Bop[k,n] with Bop row-major (n contiguous) and Aop either way.
// 128 x 128 output tile per workgroup, four waves of 64 x 64 (16 accumulator blocks), K in steps of 16
// through two LDS stages: the next step's global loads are issued before the current step's MFMAs and
// written to the other stage after them, so there is one barrier per step.  LDS rows are [k][m] / [k][n]
// with a stride of 144 floats (= 16 mod 32 banks: the four k-rows a fragment read touches land on
// disjoint banks).  K is split over blockIdx.z when M x N alone cannot fill the chip; the partial tiles
// are summed in chunk order by nw_bwd_reduce_kernel (deterministic), which also adds the rank-one term.
constexpr int GM = 128, GN = 128, GK = 16, GLD = 144;  // (GK = 32 measured: no faster, half the workgroups per CU)

template <bool A_KCONTIG, bool FUSE>
__global__ __launch_bounds__(256, 2) void nw_bwd_gemm_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm, int64_t ldb,
    float* __restrict__ Cout, const float* __restrict__ rowscale, const float* __restrict__ X, int M, int Nn,
    int K, int k_chunk) {
    __shared__ __attribute__((aligned(16))) float As[2][GK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK * GLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    const int kb = blockIdx.z * k_chunk, ke = min(K, kb + k_chunk);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NLD = GK / 8;   // float4 loads per thread and operand per step
    constexpr int KQ = GK / 4;    // float4s along k in one A row (K-contiguous form)
    float4 ra[NLD], rb[NLD];

    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int k = k0 + (tid >> 5) + 8 * u, n = n0 + 4 * (tid & 31);
            rb[u] = (k < ke && n < Nn) ? *reinterpret_cast<const float4*>(Bm + (int64_t)k * ldb + n) : zero4;
            if (A_KCONTIG) {
                const int m = m0 + tid / KQ + (256 / KQ) * u, kk = k0 + 4 * (tid % KQ);
                float4 v = zero4;
                if (m < M && kk < ke) {
                    v = *reinterpret_cast<const float4*>(A + (int64_t)m * lda + kk);
                    // the row's pad columns (k >= K) are uninitialised: 0 * NaN would poison the sum
                    if (kk + 1 >= ke) v.y = 0.f;
                    if (kk + 2 >= ke) v.z = 0.f;
                    if (kk + 3 >= ke) v.w = 0.f;
                }
                ra[u] = v;
            } else {
                const int m = m0 + 4 * (tid & 31);  // columns past M only feed output rows that are never stored
                ra[u] = (k < ke && m < M) ? *reinterpret_cast<const float4*>(A + (int64_t)k * lda + m) : zero4;
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int k = (tid >> 5) + 8 * u, c = 4 * (tid & 31);
            *reinterpret_cast<float4*>(&Bs[buf][k * GLD + c]) = rb[u];
            if (A_KCONTIG) {
                const int m = tid / KQ + (256 / KQ) * u, kk = 4 * (tid % KQ);
                As[buf][(kk + 0) * GLD + m] = ra[u].x;
                As[buf][(kk + 1) * GLD + m] = ra[u].y;
                As[buf][(kk + 2) * GLD + m] = ra[u].z;
                As[buf][(kk + 3) * GLD + m] = ra[u].w;
            } else {
                *reinterpret_cast<float4*>(&As[buf][k * GLD + c]) = ra[u];
            }
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            const int kr = (4 * kk + (lane >> 4)) * GLD + (lane & 15);
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[buf][kr + wm + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[buf][kr + wn + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nsteps = (ke - kb + GK - 1) / GK;  // >= 1: the launcher never makes an empty chunk
    gload(kb);
    sstore(0);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
        const bool more = t + 1 < nsteps;
        if (more) gload(kb + (t + 1) * GK);
        compute(t & 1);
        if (more) sstore((t + 1) & 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm + 16 * i + 4 * (lane >> 4) + r;
            if (m >= M) continue;
            const float rsc = FUSE ? 2.f * rowscale[m] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn + 16 * j + (lane & 15);
                if (n >= Nn) continue;
                if (FUSE) {
                    const int64_t o = (int64_t)m * Nn + n;
                    Cout[o] = __builtin_fmaf(rsc, X[o], acc[i][j][r]);
                } else {
                    Cout[((int64_t)blockIdx.z * M + m) * Nn + n] = acc[i][j][r];
                }
            }
        }
}

// C[m,n] = sum_chunks part[c][m,n] + 2 rowscale[m] X[m,n]   (n in float4s; Nn % 4 == 0)
__global__ __launch_bounds__(256) void nw_bwd_reduce_kernel(const float* __restrict__ part, int nchunks,
                                                             const float* __restrict__ rowscale,
                                                             const float* __restrict__ X, float* __restrict__ Cout,
                                                             int64_t M, int64_t Nn) {
    const int64_t total4 = M * Nn / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int64_t m = (i * 4) / Nn;
    float4 a = *reinterpret_cast<const float4*>(part + i * 4);
    for (int c = 1; c < nchunks; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)c * M * Nn) + i * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const float rsc = 2.f * rowscale[m];
    const float4 x = *reinterpret_cast<const float4*>(X + i * 4);
    a.x = __builtin_fmaf(rsc, x.x, a.x); a.y = __builtin_fmaf(rsc, x.y, a.y);
    a.z = __builtin_fmaf(rsc, x.z, a.z); a.w = __builtin_fmaf(rsc, x.w, a.w);
    *reinterpret_cast<float4*>(Cout + i * 4) = a;
}

// rs[j] = sum_b Rs[b*ld + j], deterministic: a workgroup owns 64 columns (16 float4 lanes: 256 contiguous bytes per
// row), its 16 row groups interleave the rows and are added in order through LDS.
__device__ __forceinline__ void colsum_block(int blk, float4 (*red)[16], const float* __restrict__ Rs, int64_t ld,
                                             float* __restrict__ rs, int64_t B, int64_t N) {
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int64_t j = ((int64_t)blk * 16 + c) * 4;   // ld % 4 == 0: whole float4s, pad columns included
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a;
    if (j < ld) {
        int64_t b = rg;
        for (; b + 16 < B; b += 32) {   // two independent loads in flight
            const float4 v = *reinterpret_cast<const float4*>(Rs + b * ld + j);
            const float4 w = *reinterpret_cast<const float4*>(Rs + (b + 16) * ld + j);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            a2.x += w.x; a2.y += w.y; a2.z += w.z; a2.w += w.w;
        }
        if (b < B) {
            const float4 v = *reinterpret_cast<const float4*>(Rs + b * ld + j);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
    }
    red[rg][c] = a;
    __syncthreads();
    if (rg == 0 && j < N) {
        for (int r = 1; r < 16; ++r) {
            const float4 v = red[r][c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        const float o[4] = {a.x, a.y, a.z, a.w};
        for (int e = 0; e < 4; ++e)
            if (j + e < N) rs[j + e] = o[e];
    }
}

// q'' = q 2^(G - E_b) as split rows, rows B .. Bpad-1 zero.  qv[b] = max_k |q[b,k]| 2^-E_b (coefficient kernel);
// every workgroup takes the maximum of qv for G (B floats from L2), workgroup 0 publishes 2^-G.
__device__ __forceinline__ void bwd_qsplit_block(int blk, float* red, const float* __restrict__ q,
                                                 const float* __restrict__ ascale, const float* __restrict__ qv,
                                                 float* __restrict__ out, float* __restrict__ gfac, int64_t B, int64_t Bpad,
                                                 int64_t d) {
    float vm = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 256) vm = fmaxf(vm, qv[b]);
    vm = block_max(vm, red);
    const int G = split_exponent(vm);
    if (blk == 0 && threadIdx.x == 0) *gfac = __builtin_ldexpf(1.f, -G);
    const int64_t r = (int64_t)blk * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= Bpad) return;
    _Float16* dst = reinterpret_cast<_Float16*>(out + r * d);
    const int64_t n4 = d / 4;
    float up = 0.f;
    if (r < B) {
        // 2^(G - E_b): both factors are powers of two; the product may leave the normal range only for rows whose
        // coefficients are 2^-100 of the batch's largest -- they contribute nothing either way
        up = ascale[r] * __builtin_ldexpf(1.f, G);
    }
    const float4* src = reinterpret_cast<const float4*>(q + (r < B ? r : 0) * d);
    for (int64_t c = lane; c < n4; c += 64) {
        const float4 v = src[c];
        const float sv[4] = {v.x * up, v.y * up, v.z * up, v.w * up};
        typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
        halfx4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (_Float16)sv[k];
            l[k] = (_Float16)(sv[k] - (float)h[k]);
        }
        const int64_t chunk = c >> 3, within = (c & 7) * 4;
        *reinterpret_cast<halfx4*>(dst + chunk * 64 + within) = h;
        *reinterpret_cast<halfx4*>(dst + chunk * 64 + 32 + within) = l;
    }
}


// q'' and rs in ONE launch (split path): both only need the coefficient kernel's outputs, neither is long enough to be
// worth a launch of its own.  Workgroups [0, nq): q''; the rest: column sums.
__global__ __launch_bounds__(256) void nw_bwd_prep_kernel(int nq, const float* __restrict__ q, const float* __restrict__ ascale,
                                                           const float* __restrict__ qv, float* __restrict__ q_split,
                                                           float* __restrict__ gfac, int64_t B, int64_t Bpad, int64_t d,
                                                           const float* __restrict__ Rs, int64_t ld, float* __restrict__ rs,
                                                           int64_t N) {
    __shared__ float4 red[16][16];
    if ((int)blockIdx.x < nq) bwd_qsplit_block(blockIdx.x, reinterpret_cast<float*>(red), q, ascale, qv, q_split, gfac, B, Bpad, d);
    else colsum_block(blockIdx.x - nq, red, Rs, ld, rs, B, N);
}
__global__ __launch_bounds__(256) void nw_colsum_kernel(const float* __restrict__ Rs, int64_t ld, float* __restrict__ rs,
                                                         int64_t B, int64_t N) {
    __shared__ float4 red[16][16];
    colsum_block(blockIdx.x, red, Rs, ld, rs, B, N);
}

struct GemmPlan {
    int nchunks, k_chunk;
};
// Split K until the grid has >= 512 workgroups (two per CU: the kernel hides its global-load latency with
// co-resident workgroups), never below 64 of K per chunk.  Measured at M=10000, N=512, K=256 (316 tiles):
// 83 us unsplit, 46 + 16 (reduce) split in two, 53 + 26 split in four.
GemmPlan gemm_plan(int64_t M, int64_t Nn, int64_t K) {
    const int64_t tiles = ((M + GM - 1) / GM) * ((Nn + GN - 1) / GN);
    int64_t want = tiles >= 512 ? 1 : (512 + tiles - 1) / tiles;
    const int64_t maxc = K / 64 > 1 ? K / 64 : 1;
    if (want > maxc) want = maxc;
    int64_t kc = (K + want - 1) / want;
    kc = (kc + GK - 1) / GK * GK;
    GemmPlan p;
    p.k_chunk = (int)kc;
    p.nchunks = (int)((K + kc - 1) / kc);
    return p;
}

template <bool A_KCONTIG>
int launch_bwd_gemm(const float* A, int64_t lda, const float* Bm, float* part, const float* rowscale, const float* X,
                    float* Cout, int64_t M, int64_t Nn, int64_t K, hipStream_t st) {
    const GemmPlan p = gemm_plan(M, Nn, K);
    const dim3 grid((unsigned)((Nn + GN - 1) / GN), (unsigned)((M + GM - 1) / GM), (unsigned)p.nchunks);
    if (grid.y > 65535u || grid.z > 65535u) return NW_ERR_INVALID_ARG;
    if (p.nchunks == 1) {
        hipLaunchKernelGGL((nw_bwd_gemm_kernel<A_KCONTIG, true>), grid, dim3(256), 0, st, A, lda, Bm, Nn, Cout, rowscale,
                           X, (int)M, (int)Nn, (int)K, p.k_chunk);
    } else {
        hipLaunchKernelGGL((nw_bwd_gemm_kernel<A_KCONTIG, false>), grid, dim3(256), 0, st, A, lda, Bm, Nn, part,
                           rowscale, X, (int)M, (int)Nn, (int)K, p.k_chunk);
        const int64_t total4 = M * Nn / 4;
        hipLaunchKernelGGL(nw_bwd_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, part,
                           p.nchunks, rowscale, X, Cout, M, Nn);
    }
    return NW_OK;
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
    float *A, *Rs, *rq, *gls, *qn2, *sn2, *rs, *part, *part2;   // part2: the second product's partial tiles (split path)
    float *s_split, *s_scale, *q_split, *ascale, *qv, *gfac;   // split path (bwd_split.hip)
    int64_t ld;    // row stride of A and Rs: N, N rounded up to 4 floats (fp32 matrix cores) or to 32 (split path)
    int64_t Bpad;  // rows of q_split: B rounded up to 32, the rest zero
    bool mfma, split;
};
// Shared support, float4-able rows and enough work to fill the chip: the two products run on the matrix cores.
bool bwd_use_mfma(int64_t B, int64_t N, int64_t d, int sup_batched) {
    const bool off = knob(KNOB_BWD_NO_MFMA) == 1;
    return !off && !sup_batched && d % 4 == 0 && B * N * d >= (int64_t)1 << 22;
}
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// The products on the fp16 matrix cores (bwd_split.hip): split rows need d % 32 == 0, the coefficient kernel stages a
// row of A (N rounded up to 32 floats) next to the C class gradients in LDS, and the extra passes (splitting the
// supports and the queries) have to pay: at B=256, N=10000, d=512 the backward's kernels take 64 us against 157 on the
// fp32 cores.
bool bwd_use_split(int64_t B, int64_t N, int64_t d, int64_t C, int sup_batched) {
    const int mode = knob(KNOB_BWD_SPLIT) == KNOB_UNSET ? -1 : knob(KNOB_BWD_SPLIT);   // 0 off, 1 wherever possible
    if (mode == 0 || !bwd_use_mfma(B, N, d, sup_batched) || d % 32 != 0) return false;
    const int64_t ld = (N + 31) / 32 * 32;
    if ((size_t)(80 + ld + C) * sizeof(float) > 150 * 1024) return false;
    if (mode == 1) return true;
    return B >= 16 && N >= 256;   // (with bwd_use_mfma's B N d >= 2^22: ahead at every such shape measured, tools/bwd_crossover.py)
}

size_t bwd_layout(int64_t B, int64_t N, int64_t d, int64_t C, int sup_batched, char* base, BwdWs* ws) {
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += align256(nfloat * sizeof(float));
        return p;
    };
    BwdWs w;
    w.mfma = bwd_use_mfma(B, N, d, sup_batched);
    w.split = bwd_use_split(B, N, d, C, sup_batched);
    w.ld = w.split ? (N + 31) / 32 * 32 : w.mfma ? (N + 3) / 4 * 4 : N;
    w.Bpad = (B + 31) / 32 * 32;
    w.A = take((size_t)B * w.ld + (w.split ? XGEMM_TAIL_BYTES / 4 : 0));
    w.Rs = take((size_t)B * w.ld);
    w.rq = take((size_t)B);
    w.gls = take((size_t)B);
    w.qn2 = take((size_t)B);
    w.sn2 = take(sup_batched ? (size_t)B * N : (size_t)N);
    w.rs = w.part = w.part2 = nullptr;
    w.s_split = w.s_scale = w.q_split = w.ascale = w.qv = w.gfac = nullptr;
    if (w.mfma) {
        w.rs = take((size_t)N);
        int cq, cs;
        if (w.split) {
            cq = xgemm_plan(B, d, N).nchunks, cs = xgemm_plan(N, d, B).nchunks;
            w.s_split = take((size_t)N * d + XGEMM_TAIL_BYTES / 4);
            w.s_scale = take((size_t)N);
            w.q_split = take((size_t)w.Bpad * d + XGEMM_TAIL_BYTES / 4);
            w.ascale = take((size_t)B);
            w.qv = take((size_t)B);
            w.gfac = take(1);
        } else {
            cq = gemm_plan(B, d, N).nchunks, cs = gemm_plan(N, d, B).nchunks;
        }
        const size_t nq = cq > 1 ? (size_t)cq * B * d : 0, ns = cs > 1 ? (size_t)cs * N * d : 0;
        if (w.split) {   // the first product's partial tiles are still pending while the second product runs
            w.part = take(nq);
            w.part2 = take(ns);
        } else {
            w.part = w.part2 = take(nq > ns ? nq : ns);
        }
    }
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
    (void)kind;
    if (B <= 0 || N < 0) return 0;
    return nw::bwd_layout(B, N, d, C, sup_batched, nullptr, nullptr);
}

extern "C" int nw_bwd_uses_split(int64_t B, int64_t N, int64_t d, int64_t C, int sup_batched) {
    if (B <= 0 || N <= 0 || d <= 0 || C < 0) return 0;
    return nw::bwd_use_split(B, N, d, C, sup_batched) ? 1 : 0;
}

extern "C" int nw_bwd_f32(const float* q, const float* s, const int64_t* sy, const float* scores,
                          const float* lse, const float* out, const float* gout, float* gq,
                          float* gs, float* glogit_scale, void* workspace, size_t workspace_bytes,
                          int64_t B, int64_t N, int64_t d, int64_t C, int kind,
                          const float* logit_scale_dev, int sup_batched, int labels_batched,
                          void* stream) {
    return nw_bwd_bank_f32(q, s, nullptr, nullptr, nullptr, sy, scores, lse, out, gout, gq, gs, glogit_scale, workspace,
                           workspace_bytes, B, N, d, C, kind, logit_scale_dev, sup_batched, labels_batched, stream);
}

extern "C" int nw_bwd_bank_f32(const float* q, const float* s, const float* s_norm2, const float* s_split,
                               const float* s_scale, const int64_t* sy, const float* scores,
                               const float* lse, const float* out, const float* gout, float* gq,
                               float* gs, float* glogit_scale, void* workspace, size_t workspace_bytes,
                               int64_t B, int64_t N, int64_t d, int64_t C, int kind,
                               const float* logit_scale_dev, int sup_batched, int labels_batched,
                               void* stream) {
    using namespace nw;
    if ((s_split != nullptr) != (s_scale != nullptr) || (s_split && !s_norm2)) return NW_ERR_INVALID_ARG;
    if (s_split && (sup_batched || d % 32 != 0 || (reinterpret_cast<uintptr_t>(s_split) & 15))) return NW_ERR_INVALID_ARG;
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
    const size_t need = bwd_layout(B, N, d, C, sup_batched, static_cast<char*>(workspace), &ws);
    if (!workspace || workspace_bytes < need) return NW_ERR_WORKSPACE;
    const size_t lds = (80 + (size_t)C + (ws.split ? (size_t)ws.ld : 0)) * sizeof(float);
    const int coeff_env = knob(KNOB_COEFF_THREADS);   // timing experiments
    const unsigned coeff_threads = (coeff_env == 256 || coeff_env == 512 || coeff_env == 1024) ? (unsigned)coeff_env : (N >= 2048 ? 1024 : 256);
    if (lds > 160 * 1024) return NW_ERR_UNSUPPORTED;
    const bool aligned = ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(s) |
                           reinterpret_cast<uintptr_t>(gq) | reinterpret_cast<uintptr_t>(gs)) & 15) == 0;
    if (ws.mfma && !aligned) return NW_ERR_INVALID_ARG;  // the layout (row stride of A) is already the matrix-core one

    const bool norms = (kind == NW_SCORE_HYPERSPHERE || kind == NW_SCORE_COSINE || kind == NW_SCORE_CLIP);
    if (norms) hipLaunchKernelGGL(nw_rownorm_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, q, ws.qn2, B, d);
    // The caller's bank is the Y operand of a 64-column tile kernel that reads a clamped last row to the end of its
    // tile: only a row length that is a multiple of 64 keeps it inside N * d floats (the workspace copy has a tail)
    if (ws.split && s_split && d % 64 == 0) {   // the forward's bank: split rows, scales and norms of these very supports
        ws.s_split = const_cast<float*>(s_split);
        ws.s_scale = const_cast<float*>(s_scale);
        ws.sn2 = const_cast<float*>(s_norm2);
    } else if (ws.split) {   // the supports as split rows (the bank format); their squared norms come with it
        const int rc = launch_split_rows(s, ws.s_split, ws.s_scale, ws.sn2, N, d, st);
        if (rc != NW_OK) return rc;
    } else if (norms && s_norm2 && !sup_batched) {
        ws.sn2 = const_cast<float*>(s_norm2);
    } else if (norms) {
        hipLaunchKernelGGL(nw_rownorm_kernel, dim3((unsigned)((gs_rows + 3) / 4)), dim3(256), 0, st, s, ws.sn2, gs_rows, d);
    }
#define NW_COEFF(K)                                                                                                      \
    do {                                                                                                                 \
        if (ws.split)                                                                                                    \
            hipLaunchKernelGGL((nw_bwd_coeff_kernel<K, true>), dim3((unsigned)B), dim3(coeff_threads), lds, st, scores, lse, \
                               out, gout, sy, labels_batched, ws.qn2, ws.sn2, sup_batched, logit_scale_dev, ws.A, ws.Rs,  \
                               ws.rq, ws.gls, N, C, ws.ld, q, d, ws.s_scale, ws.ascale, ws.qv);                          \
        else                                                                                                             \
            hipLaunchKernelGGL((nw_bwd_coeff_kernel<K, false>), dim3((unsigned)B), dim3(coeff_threads), lds, st, scores, lse, \
                               out, gout, sy, labels_batched, ws.qn2, ws.sn2, sup_batched, logit_scale_dev, ws.A, ws.Rs,  \
                               ws.rq, ws.gls, N, C, ws.ld);                                                              \
    } while (0)
    switch (kind) {
        case NW_SCORE_EUCLIDEAN: NW_COEFF(NW_SCORE_EUCLIDEAN); break;
        case NW_SCORE_HYPERSPHERE: NW_COEFF(NW_SCORE_HYPERSPHERE); break;
        case NW_SCORE_COSINE: NW_COEFF(NW_SCORE_COSINE); break;
        case NW_SCORE_DOT: NW_COEFF(NW_SCORE_DOT); break;
        default: NW_COEFF(NW_SCORE_CLIP); break;
    }
#undef NW_COEFF
    if (ws.mfma) {
        const unsigned ncs = (unsigned)((ws.ld / 4 + 15) / 16);
        if (ws.split) {
            const unsigned nq = (unsigned)((ws.Bpad + 3) / 4);
            hipLaunchKernelGGL(nw_bwd_prep_kernel, dim3(nq + ncs), dim3(256), 0, st, (int)nq, q, ws.ascale, ws.qv, ws.q_split,
                               ws.gfac, B, ws.Bpad, d, ws.Rs, ws.ld, ws.rs, N);
        } else {
            hipLaunchKernelGGL(nw_colsum_kernel, dim3(ncs), dim3(256), 0, st, ws.Rs, ws.ld, ws.rs, B, N);
        }
        int rc;
        if (ws.split) {
            // gq = 2^-E_b (A' s') + 2 rq q: K = N runs along the rows of A' (zero past N), the rows of s' are clamped
            XgemmReduce gq_reduce;   // the first product's K-split reduction rides along with the second product's launch
            rc = launch_xgemm(false, ws.A, ws.ld, B, ws.s_split, d, N, ws.part, ws.ascale, 0, nullptr, ws.rq, q, gq, B, d,
                              N, st, &gq_reduce);
            if (rc != NW_OK) return rc;
            // gs = 2^(e_j - G) (A'^T q'') + 2 rs s: K = B runs across the rows of A' (clamped), q'' is zero past B
            rc = launch_xgemm(true, ws.A, ws.ld, B, ws.q_split, d, ws.Bpad, ws.part2, ws.s_scale, 1, ws.gfac, ws.rs, s, gs,
                              N, d, B, st, nullptr, &gq_reduce);
            if (rc != NW_OK) return rc;
        } else {
            rc = launch_bwd_gemm<true>(ws.A, ws.ld, s, ws.part, ws.rq, q, gq, B, d, N, st);   // gq = A s + 2 rq q
            if (rc != NW_OK) return rc;
            rc = launch_bwd_gemm<false>(ws.A, ws.ld, q, ws.part, ws.rs, s, gs, N, d, B, st);      // gs = A^T q + 2 rs s
            if (rc != NW_OK) return rc;
        }
        if (glogit_scale) {
            if (kind == NW_SCORE_CLIP)
                hipLaunchKernelGGL(nw_sum_kernel, dim3(1), dim3(256), 0, st, ws.gls, glogit_scale, B);
            else if (hipMemsetAsync(glogit_scale, 0, 4, st) != hipSuccess)
                return NW_ERR_LAUNCH;
        }
        NW_CHECK_LAUNCH();
        return NW_OK;
    }
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

// gradient of the aggregation alone w.r.t. a given score matrix: the DOT-product form of the coefficient kernel
// (A = dS) writes it straight into gscores
extern "C" int nw_aggregate_bwd_f32(const float* scores, const int64_t* sy, const float* lse, const float* out,
                                    const float* gout, float* gscores, int64_t B, int64_t N, int64_t C,
                                    int labels_batched, void* stream) {
    using namespace nw;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B < 0 || N < 0 || C < 0) return NW_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return NW_OK;
    if (!scores || !sy || !lse || !out || !gout || !gscores) return NW_ERR_INVALID_ARG;
    if (B > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    const size_t lds = (80 + (size_t)C) * sizeof(float);
    if (lds > 160 * 1024) return NW_ERR_UNSUPPORTED;
    const unsigned threads = N >= 2048 ? 1024 : 256;
    hipLaunchKernelGGL(nw_bwd_coeff_kernel<NW_SCORE_DOT>, dim3((unsigned)B), dim3(threads), lds, st, scores, lse, out, gout,
                       sy, labels_batched, (const float*)nullptr, (const float*)nullptr, 0, (const float*)nullptr, gscores,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, N, C, N);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
