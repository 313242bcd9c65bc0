// bn_nhwc.hip -- training-mode BatchNorm2d (+ ReLU) over channels-last activations, forward and backward (gfx950 /
// MI355X only).  The NHWC counterpart of bnrelu.hip's one-workgroup-per-channel kernels, for the training path whose
// convolutions run in csrc/conv_nhwc.hip: same call sites (model/densenet.py:33-60 norm1 / norm2, :82-91, :139;
// model/resnet.py:31-66), same arithmetic (shifted / merged moments, centred normalisation, the ReLU mask recomputed
// from x in the backward), and every pass that writes an activation also leaves its amax record (per-workgroup maxima,
// include/nwhead_hip.h) for the convolution that reads it next.
//
// A tensor is (rows = n h w, C) with a row stride ldx >= C (a channel prefix of a wider NHWC tensor qualifies),
// C % 4 == 0.  Three launches each way (the middle one tiny):
//   forward : stats (per row chunk and channel: count, mean, M2 -- Welford moments of the chunk, merged pairwise in a
//             fixed order with Chan's formula: no E[x^2] - E[x]^2 cancellation, deterministic)
//             -> finalize (64 channels x 16 chunk lanes per workgroup merge the chunks: mean, 1/sqrt(var + eps), running
//                statistics, step counter; ~3 us -- merged by every apply workgroup instead it cost 9 us per pass)
//             -> apply (y = max((x - mean) a + beta, 0), a = gamma invstd; amax record of y)
//   backward: stats (per chunk and channel: sum g, sum g xhat, g = dy [y > 0]) -> finalize (dgamma, dbeta, the two
//             means) -> apply (dx = a (g - mean(g) - xhat mean(g xhat)) [+ acc]; amax record of dx)
#include "nw_internal.h"
#include <cstdlib>

namespace nw {
namespace {

constexpr int BN_SLOTS = 256;        // = NW_AMAX_SLOTS
constexpr int BN_MAX_C = 2560;       // channels: the backward apply keeps six per-channel factors in LDS (60 KB; DenseNet-161 / -201
                                     // reach 2208 / 1920 channels in front of their last BatchNorm)
constexpr int BN_TC = 64;            // channels per stats workgroup: 16 float4 lanes x 64 row lanes
constexpr int BN_RL = 64;            // row lanes of a stats workgroup (1024 threads)
constexpr int BN_GC = 16384;         // chunks x channels of the partial moments (196 KB: what every apply workgroup re-reads)
constexpr int BN_INLINE_ROWS = 16384;  // backward of maps this small: no finalize launch, the apply workgroups merge
constexpr int BN_INLINE_GC = 8192;     //   at most this many chunk sums themselves

__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float n2, float mean2, float m22) {
    const float nt = n + n2;
    if (n2 > 0.f) {
        const float d = mean2 - mean, f = n2 / nt;
        mean = __builtin_fmaf(d, f, mean);
        m2 = m2 + m22 + d * d * n * f;
    }
    n = nt;
}

// partial moments of rows [r0, r1) for the 64 channels of tile blockIdx.y: part[(k * G + g) * C + c], k = 0 count, 1 mean, 2 M2
// mm != 0: also k = 3 minimum, 4 maximum of the chunk (what a consumer that applies BatchNorm + ReLU on the fly needs to bound it)
__global__ __launch_bounds__(1024) void nw_bn_nhwc_stats_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ part,
                                                                 int64_t R, int C, int G, int64_t rows_per_chunk, int mm) {
    __shared__ float sh[3][BN_RL][BN_TC + 1];
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int c0 = blockIdx.y * BN_TC + 4 * cq;
    const int g = blockIdx.x;
    const int64_t r0 = (int64_t)g * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float n = 0.f, K[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float vlo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, vhi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (c0 < C) {
        int64_t r = r0 + rl;
        if (r < r1) {   // the shift: this thread's first value per channel
            const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c0);
            K[0] = v.x; K[1] = v.y; K[2] = v.z; K[3] = v.w;
        }
        for (; r < r1; r += BN_RL) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c0);
            const float d[4] = {v.x - K[0], v.y - K[1], v.z - K[2], v.w - K[3]};
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s1[j] += d[j];
                s2[j] = __builtin_fmaf(d[j], d[j], s2[j]);
                vlo[j] = fminf(vlo[j], vv[j]);
                vhi[j] = fmaxf(vhi[j], vv[j]);
            }
            n += 1.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float md = n > 0.f ? s1[j] / n : 0.f;
        sh[0][rl][4 * cq + j] = n;
        sh[1][rl][4 * cq + j] = K[j] + md;
        sh[2][rl][4 * cq + j] = n > 0.f ? fmaxf(s2[j] - md * s1[j], 0.f) : 0.f;
    }
    __syncthreads();
    // row lanes merged in a fixed order: 4 x 16 sequential merges, then the four partial results
    const int c = tid & 63, j4 = tid >> 6;
    float nn = 0.f, mean = 0.f, m2 = 0.f;
    if (tid < 256)
        for (int l = 16 * j4; l < 16 * j4 + 16; ++l) chan_merge(nn, mean, m2, sh[0][l][c], sh[1][l][c], sh[2][l][c]);
    __syncthreads();
    if (tid < 256) { sh[0][j4][c] = nn; sh[1][j4][c] = mean; sh[2][j4][c] = m2; }
    __syncthreads();
    if (tid < BN_TC && blockIdx.y * BN_TC + tid < C) {
        nn = sh[0][0][tid]; mean = sh[1][0][tid]; m2 = sh[2][0][tid];
        for (int l = 1; l < 4; ++l) chan_merge(nn, mean, m2, sh[0][l][tid], sh[1][l][tid], sh[2][l][tid]);
        const int cc = blockIdx.y * BN_TC + tid;
        part[((int64_t)0 * G + g) * C + cc] = nn;
        part[((int64_t)1 * G + g) * C + cc] = mean;
        part[((int64_t)2 * G + g) * C + cc] = m2;
    }
    if (!mm) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[0][rl][4 * cq + j] = vlo[j]; sh[1][rl][4 * cq + j] = vhi[j]; }
    __syncthreads();
    if (tid < BN_TC && blockIdx.y * BN_TC + tid < C) {
        float lo = INFINITY, hi = -INFINITY;
        for (int l = 0; l < BN_RL; ++l) { lo = fminf(lo, sh[0][l][tid]); hi = fmaxf(hi, sh[1][l][tid]); }
        const int cc = blockIdx.y * BN_TC + tid;
        part[((int64_t)3 * G + g) * C + cc] = lo;
        part[((int64_t)4 * G + g) * C + cc] = hi;
    }
}

// Merge of the G chunk moments: 64 channels x 16 chunk lanes per workgroup -- lane j merges the chunks j, j + 16, ... of
// its channel (coalesced across channels, <= 16 dependent steps), the 16 partial results are merged in lane order.
__global__ __launch_bounds__(1024) void nw_bn_nhwc_finalize_kernel(const float* __restrict__ part, int G, int C,
                                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                    int64_t* __restrict__ num_batches_tracked, float momentum, float eps,
                                                                    float* __restrict__ save_var, float* __restrict__ save_min = nullptr,
                                                                    float* __restrict__ save_max = nullptr) {
    __shared__ float sh[3][16][64];
    __shared__ float shm[2][16][64];
    const int cl = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (c < C)
        for (int g = j; g < G; g += 16)
            chan_merge(n, mean, m2, part[(int64_t)g * C + c], part[((int64_t)G + g) * C + c], part[((int64_t)2 * G + g) * C + c]);
    sh[0][j][cl] = n; sh[1][j][cl] = mean; sh[2][j][cl] = m2;
    if (save_min) {   // (the statistics pass left rows 3 and 4: nw_bn_nhwc_stats_kernel's mm)
        float lo = INFINITY, hi = -INFINITY;
        if (c < C)
            for (int g = j; g < G; g += 16) {
                lo = fminf(lo, part[((int64_t)3 * G + g) * C + c]);
                hi = fmaxf(hi, part[((int64_t)4 * G + g) * C + c]);
            }
        shm[0][j][cl] = lo; shm[1][j][cl] = hi;
    }
    __syncthreads();
    if (j != 0 || c >= C) return;
    for (int k = 1; k < 16; ++k) chan_merge(n, mean, m2, sh[0][k][cl], sh[1][k][cl], sh[2][k][cl]);
    if (save_min) {
        float lo = shm[0][0][cl], hi = shm[1][0][cl];
        for (int k = 1; k < 16; ++k) { lo = fminf(lo, shm[0][k][cl]); hi = fmaxf(hi, shm[1][k][cl]); }
        save_min[c] = lo;
        save_max[c] = hi;
    }
    const float var = n > 0.f ? fmaxf(m2 / n, 0.f) : 0.f;
    save_mean[c] = mean;
    save_invstd[c] = 1.f / sqrtf(var + eps);
    if (save_var) save_var[c] = var;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
}

// The moments of MANY groups (what a convolution's epilogue leaves: up to thousands of 16..64-pixel groups) to mean,
// 1/sqrt(var + eps) and the biased variance, in parallel over the groups: FOUR channels (one float4) x 256 group lanes per
// workgroup, two passes over the partials -- the weighted mean, then sum(M2_g + n_g (mean_g - mean)^2) -- each summed per
// lane over groups j, j + 256, ... and over the lanes in a fixed order (deterministic).  (Chan's pairwise merge of
// nw_bn_nhwc_finalize_kernel is a serial chain per lane: fine for <= 256 chunks, 130 dependent steps at 2058 groups; a
// first version with 16 channels x 64 lanes per workgroup ran on C / 16 = 2..8 CUs: 26 us on the 56 x 56 layers.)
// Round 4: the groups may carry minima and maxima too (rows 3, 4: save_min / save_max), and the BatchNorm that reads this tensor
// next can be PREPARED here (prep.tab != nullptr): its per-channel table mean | a | beta for the convolution that applies it in
// its loaders (conv_nhwc.hip's ConvP::pre), the exact bound on |act((x - mean) a + beta)| as slot blockIdx.x of an amax record,
// its running statistics and step counter.
struct BnPrep {
    const float* gamma; const float* beta;
    float* tab;              // [3][C]
    float* amax;             // BN_SLOTS floats
    float* running_mean; float* running_var; int64_t* num_batches_tracked;
    float momentum; int relu;
};
// bound of act((x - mean) a + beta) over x in [lo, hi]: the map is monotone in x, and computed exactly as the loaders compute it
__device__ __forceinline__ float prep_bound(float lo, float hi, float mean, float a, float b, int relu) {
    const float u = __builtin_fmaf(lo - mean, a, b), v = __builtin_fmaf(hi - mean, a, b);
    if (!(u == u) || !(v == v)) return INFINITY;   // a NaN in the statistics: the consumer turns a non-finite bound into NaN outputs
    return relu ? fmaxf(fmaxf(u, v), 0.f) : fmaxf(fabsf(u), fabsf(v));
}
__global__ __launch_bounds__(1024) void nw_bn_nhwc_merge_groups_kernel(const float* __restrict__ part, int G, int C, float eps,
                                                                        float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                        float* __restrict__ save_var, float* __restrict__ save_min,
                                                                        float* __restrict__ save_max, const BnPrep prep) {
    __shared__ float4 sh[1024];
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * 4;                               // this workgroup's four channels (C % 4 == 0)
    auto add4 = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    auto block_sum4 = [&](float4 v) {                            // fixed-order tree over the 1024 threads; result in all
        __syncthreads();
        sh[t] = v;
        __syncthreads();
        for (int w = 512; w > 0; w >>= 1) {
            if (t < w) sh[t] = add4(sh[t], sh[t + w]);
            __syncthreads();
        }
        return sh[0];
    };
    float4 sn = make_float4(0.f, 0.f, 0.f, 0.f), sm = sn;
    for (int g = t; g < G; g += 1024) {
        const float4 n = *reinterpret_cast<const float4*>(part + (int64_t)g * C + c0);
        const float4 m = *reinterpret_cast<const float4*>(part + ((int64_t)G + g) * C + c0);
        sn = add4(sn, n);
        sm.x = __builtin_fmaf(n.x, m.x, sm.x); sm.y = __builtin_fmaf(n.y, m.y, sm.y);
        sm.z = __builtin_fmaf(n.z, m.z, sm.z); sm.w = __builtin_fmaf(n.w, m.w, sm.w);
    }
    const float4 tn = block_sum4(sn);
    const float4 tm = block_sum4(sm);
    const float4 mean = make_float4(tn.x > 0.f ? tm.x / tn.x : 0.f, tn.y > 0.f ? tm.y / tn.y : 0.f, tn.z > 0.f ? tm.z / tn.z : 0.f,
                                    tn.w > 0.f ? tm.w / tn.w : 0.f);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = t; g < G; g += 1024) {
        const float4 n = *reinterpret_cast<const float4*>(part + (int64_t)g * C + c0);
        const float4 m = *reinterpret_cast<const float4*>(part + ((int64_t)G + g) * C + c0);
        const float4 m2 = *reinterpret_cast<const float4*>(part + ((int64_t)2 * G + g) * C + c0);
        const float dx = m.x - mean.x, dy = m.y - mean.y, dz = m.z - mean.z, dw = m.w - mean.w;
        q.x += __builtin_fmaf(n.x * dx, dx, m2.x); q.y += __builtin_fmaf(n.y * dy, dy, m2.y);
        q.z += __builtin_fmaf(n.z * dz, dz, m2.z); q.w += __builtin_fmaf(n.w * dw, dw, m2.w);
    }
    const float4 tq = block_sum4(q);
    float4 tlo = make_float4(INFINITY, INFINITY, INFINITY, INFINITY), thi = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (save_min) {
        for (int g = t; g < G; g += 1024) {
            const float4 a = *reinterpret_cast<const float4*>(part + ((int64_t)3 * G + g) * C + c0);
            const float4 b = *reinterpret_cast<const float4*>(part + ((int64_t)4 * G + g) * C + c0);
            tlo = make_float4(fminf(tlo.x, a.x), fminf(tlo.y, a.y), fminf(tlo.z, a.z), fminf(tlo.w, a.w));
            thi = make_float4(fmaxf(thi.x, b.x), fmaxf(thi.y, b.y), fmaxf(thi.z, b.z), fmaxf(thi.w, b.w));
        }
        auto block_ext4 = [&](float4 v, bool mx) {
            __syncthreads();
            sh[t] = v;
            __syncthreads();
            for (int w = 512; w > 0; w >>= 1) {
                if (t < w) {
                    const float4 a = sh[t], b = sh[t + w];
                    sh[t] = mx ? make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w))
                               : make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w));
                }
                __syncthreads();
            }
            return sh[0];
        };
        tlo = block_ext4(tlo, false);
        thi = block_ext4(thi, true);
    }
    if (t != 0) return;
    const float mn[4] = {mean.x, mean.y, mean.z, mean.w}, nn[4] = {tn.x, tn.y, tn.z, tn.w}, qq[4] = {tq.x, tq.y, tq.z, tq.w};
    const float lo4[4] = {tlo.x, tlo.y, tlo.z, tlo.w}, hi4[4] = {thi.x, thi.y, thi.z, thi.w};
    float bound = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float var = nn[k] > 0.f ? fmaxf(qq[k] / nn[k], 0.f) : 0.f;
        const float inv = 1.f / sqrtf(var + eps);
        save_mean[c0 + k] = mn[k];
        save_invstd[c0 + k] = inv;
        save_var[c0 + k] = var;
        if (save_min) { save_min[c0 + k] = lo4[k]; save_max[c0 + k] = hi4[k]; }
        if (prep.tab) {
            const int c = c0 + k;
            const float a = prep.gamma[c] * inv, b = prep.beta[c];
            prep.tab[c] = mn[k];
            prep.tab[C + c] = a;
            prep.tab[2 * C + c] = b;
            bound = fmaxf(bound, prep_bound(lo4[k], hi4[k], mn[k], a, b, prep.relu));
            if (prep.running_mean) prep.running_mean[c] = (1.f - prep.momentum) * prep.running_mean[c] + prep.momentum * mn[k];
            if (prep.running_var)
                prep.running_var[c] = (1.f - prep.momentum) * prep.running_var[c] + prep.momentum * var * (nn[k] > 1.f ? nn[k] / (nn[k] - 1.f) : 1.f);
        }
    }
    if (prep.tab) {
        if (prep.num_batches_tracked && blockIdx.x == 0) *prep.num_batches_tracked += 1;
        if (prep.amax)
            for (int k = blockIdx.x; k < BN_SLOTS; k += gridDim.x) prep.amax[k] = k == (int)blockIdx.x ? bound : 0.f;
    }
}

// The same preparation from statistics that exist already (a dense block's slab channels: one layer's norm1 after another
// reads them, each with its own gamma / beta): 256 channels per workgroup.
__global__ __launch_bounds__(256) void nw_bn_nhwc_prep_kernel(const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ var, const float* __restrict__ vmin,
                                                              const float* __restrict__ vmax, int C, float unbias, const BnPrep prep) {
    __shared__ float red[8];
    const int c = blockIdx.x * 256 + threadIdx.x;
    float bound = 0.f;
    if (c < C) {
        const float m = mean[c], a = prep.gamma[c] * invstd[c], b = prep.beta[c];
        prep.tab[c] = m;
        prep.tab[C + c] = a;
        prep.tab[2 * C + c] = b;
        bound = prep_bound(vmin[c], vmax[c], m, a, b, prep.relu);
        if (prep.running_mean) prep.running_mean[c] = (1.f - prep.momentum) * prep.running_mean[c] + prep.momentum * m;
        if (prep.running_var) prep.running_var[c] = (1.f - prep.momentum) * prep.running_var[c] + prep.momentum * var[c] * unbias;
    }
    if (prep.num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *prep.num_batches_tracked += 1;
    bound = block_max(bound, red);
    if (prep.amax && (int)threadIdx.x % (int)gridDim.x == (int)blockIdx.x) prep.amax[threadIdx.x] = threadIdx.x == blockIdx.x ? bound : 0.f;
}

// Merge + preparation in ONE launch of 256-thread workgroups (round 4; the first version -- nw_bn_nhwc_merge_groups_kernel with
// its five 1024-thread tree reductions over uncoalesced float4 gathers and a one-thread tail of dependent loads, then
// nw_bn_nhwc_prep_kernel -- cost 13.7 + 7.9 us per BatchNorm, 1.9 ms per K4 step):
//   workgroups [0, ceil(C / 64)): up to 64 channels each of the tensor a convolution has just written.  Thread (quad q, lane j)
//     merges the groups j, j + GL, ... of its four channels (coalesced rows; Chan's merge, minima, maxima) in ONE pass, the GL
//     lane results of a channel are merged in lane order by the thread that owns the channel -- which requested gamma, beta and
//     the running statistics at the top of the kernel;
//   workgroups behind them: channels [0, nold) whose statistics exist already, 256 per workgroup.
// The statistics go to entries [off, off + C) of the arrays (a dense block's slab-wide ones, or a tensor's own with off = 0);
// with prep.tab != nullptr every workgroup also writes its channels' entries of the NEXT BatchNorm's table (row stride tc:
// mean | a | beta) and its bound into its slots of the amax record.
struct MergeP {
    const float* part; int G, C, off;
    float eps;
    float *mean, *invstd, *var, *vmin, *vmax;     // statistics arrays (entries [off, off + C) written, [0, nold) read)
    int nold, tc;
    float unbias_old;                             // rows / (rows - 1) for the old channels' running variance
    BnPrep prep;
};
__global__ __launch_bounds__(1024) void nw_bn_nhwc_merge_prep_kernel(const MergeP p) {
    // 16 channels per workgroup of 1024 threads: thread t = 4 j + q reads float4 quad q of the groups j, j + 256, ... (a wave-
    // instruction covers 16 rows x 64 contiguous bytes; the partials are ~1/3 of the tensor the convolution wrote, so the
    // lanes in flight matter: 256 threads x 64 channels took 18 us, one wave per quad -- 64 rows per instruction -- 14 us).
    // Lane sums by shuffles inside a wave, the 16 waves' results through LDS; two passes of plain sums (Chan's pairwise merge
    // needs a division per step): the weighted mean, then sum(M2_g + n_g (mean_g - mean)^2) with the minima and maxima.
    __shared__ float sh[5][16][17];
    __shared__ float redb[16], redm[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const BnPrep& pr = p.prep;
    const int nm = (p.C + 15) >> 4;
    float bound = 0.f;
    if (pr.tab && pr.num_batches_tracked && blockIdx.x == 0 && t == 0) *pr.num_batches_tracked += 1;
    if ((int)blockIdx.x < nm) {
        const int cb = blockIdx.x * 16;
        const int nch = min(16, p.C - cb);
        const float* part = p.part;
        const int G = p.G, C = p.C;
        float f_gamma = 1.f, f_beta = 0.f, f_rm = 0.f, f_rv = 0.f;
        const float Kc_ = t < nch ? part[(int64_t)G * C + cb + t] : 0.f;   // group 0's mean of channel t: the shift (below)
        if (t < nch && pr.tab) {
            const int c = p.off + cb + t;
            f_gamma = pr.gamma[c];
            f_beta = pr.beta[c];
            if (pr.running_mean) f_rm = pr.running_mean[c];
            if (pr.running_var) f_rv = pr.running_var[c];
        }
        const int q = t & 3, j = t >> 2;
        const bool live = 4 * q < nch;
        // sums over the lanes with equal (lane & 3): two DPP row rotations inside each row of 16, then the four rows by permlane
        // swaps -- VALU only (as 80 ds_bpermute per thread the reductions alone were ~2 us of LDS traffic)
        auto ror = [](float x, auto ctl) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctl)::value, 0xf, 0xf, false));
        };
        using R8 = std::integral_constant<int, 0x128>;
        using R4 = std::integral_constant<int, 0x124>;
        auto xsum = [&](float v) { v += ror(v, R8{}); v += ror(v, R4{}); return group4_sum(v); };
        auto xmin = [&](float v) { v = fminf(v, ror(v, R8{})); v = fminf(v, ror(v, R4{})); return group4_min(v); };
        auto xmax = [&](float v) { v = fmaxf(v, ror(v, R8{})); v = fmaxf(v, ror(v, R4{})); return group4_max(v); };
        // ONE pass (the partials come from another XCD's L2 through memory: every dependent round of loads is ~2 us):
        // sums shifted by K = the mean of group 0 -- a group mean lies within a fraction of sigma of the batch mean, so
        // var = (S2 - S1^2 / N) / N loses nothing to cancellation -- with the minima and maxima in the same loop
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        float K[4] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const float4 k4 = *reinterpret_cast<const float4*>(part + (int64_t)G * C + cb + 4 * q);
            K[0] = k4.x; K[1] = k4.y; K[2] = k4.z; K[3] = k4.w;
#pragma unroll 4
            for (int g = j; g < G; g += 256) {
                const float* row = part + (int64_t)g * C + cb + 4 * q;
                const float4 a = *reinterpret_cast<const float4*>(row);
                const float4 b = *reinterpret_cast<const float4*>(row + (int64_t)G * C);
                const float4 c2 = *reinterpret_cast<const float4*>(row + (int64_t)2 * G * C);
                const float4 d = *reinterpret_cast<const float4*>(row + (int64_t)3 * G * C);
                const float4 e = *reinterpret_cast<const float4*>(row + (int64_t)4 * G * C);
                const float an[4] = {a.x, a.y, a.z, a.w}, bm[4] = {b.x, b.y, b.z, b.w}, cc[4] = {c2.x, c2.y, c2.z, c2.w};
                const float dl[4] = {d.x, d.y, d.z, d.w}, eh[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dd = bm[k] - K[k], nd = an[k] * dd;
                    s0[k] += an[k];
                    s1[k] += nd;
                    s2[k] += __builtin_fmaf(nd, dd, cc[k]);
                    lo[k] = fminf(lo[k], dl[k]);
                    hi[k] = fmaxf(hi[k], eh[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = xsum(s0[k]), b = xsum(s1[k]), c3 = xsum(s2[k]), d = xmin(lo[k]), e = xmax(hi[k]);
            if (lane < 4) { sh[0][w][4 * q + k] = a; sh[1][w][4 * q + k] = b; sh[2][w][4 * q + k] = c3; sh[3][w][4 * q + k] = d; sh[4][w][4 * q + k] = e; }
        }
        __syncthreads();
        if (t < nch) {
            float n0 = 0.f, t1 = 0.f, t2 = 0.f, lo1 = INFINITY, hi1 = -INFINITY;
            for (int l = 0; l < 16; ++l) {
                n0 += sh[0][l][t]; t1 += sh[1][l][t]; t2 += sh[2][l][t];
                lo1 = fminf(lo1, sh[3][l][t]);
                hi1 = fmaxf(hi1, sh[4][l][t]);
            }
            const float Kc = Kc_;                                  // (the same value every lane of the channel used)
            const float dm = n0 > 0.f ? t1 / n0 : 0.f;
            const float mn = Kc + dm;
            const float var = n0 > 0.f ? fmaxf((t2 - dm * t1) / n0, 0.f) : 0.f;
            const float nn = n0;
            const int c = p.off + cb + t;
            const float inv = 1.f / sqrtf(var + p.eps);
            p.mean[c] = mn; p.invstd[c] = inv; p.var[c] = var; p.vmin[c] = lo1; p.vmax[c] = hi1;
            if (pr.tab) {
                const float a = f_gamma * inv;
                pr.tab[c] = mn; pr.tab[p.tc + c] = a; pr.tab[2 * p.tc + c] = f_beta;
                bound = prep_bound(lo1, hi1, mn, a, f_beta, pr.relu);
                if (pr.running_mean) pr.running_mean[c] = (1.f - pr.momentum) * f_rm + pr.momentum * mn;
                if (pr.running_var) pr.running_var[c] = (1.f - pr.momentum) * f_rv + pr.momentum * var * (nn > 1.f ? nn / (nn - 1.f) : 1.f);
            }
        }
    } else if (pr.tab) {
        const int c = ((int)blockIdx.x - nm) * 1024 + t;
        if (c < p.nold) {
            const float m = p.mean[c], a = pr.gamma[c] * p.invstd[c], b = pr.beta[c];
            pr.tab[c] = m; pr.tab[p.tc + c] = a; pr.tab[2 * p.tc + c] = b;
            bound = prep_bound(p.vmin[c], p.vmax[c], m, a, b, pr.relu);
            if (pr.running_mean) pr.running_mean[c] = (1.f - pr.momentum) * pr.running_mean[c] + pr.momentum * m;
            if (pr.running_var) pr.running_var[c] = (1.f - pr.momentum) * pr.running_var[c] + pr.momentum * p.var[c] * p.unbias_old;
        }
    }
    if (!pr.tab) return;
    bound = block_max(bound, redb);
    // the record's slots b, b + g, ... belong to workgroup b of g: its bound in the first, zeros in the others, one store per thread
    if (pr.amax && t < BN_SLOTS && t % (int)gridDim.x == (int)blockIdx.x) pr.amax[t] = t == (int)blockIdx.x ? bound : 0.f;
}

// y[r][c] = act((x[r][c] - mean[c]) a[c] + beta[c]), a = gamma invstd; amax record of y
template <bool RELU>
__global__ __launch_bounds__(1024) void nw_bn_nhwc_apply_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                                 const float* __restrict__ save_invstd, float* __restrict__ y,
                                                                 float* __restrict__ amax, int64_t R, int C,
                                                                 const float* __restrict__ batch_var, float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked,
                                                                 float momentum, float unbias) {
    extern __shared__ float prm[];   // [3][C]: mean, a, beta
    __shared__ float red[16];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        prm[c] = save_mean[c];
        prm[C + c] = gamma[c] * save_invstd[c];
        prm[2 * C + c] = beta[c];
    }
    if (blockIdx.x == 0 && batch_var) {   // statistics that came from elsewhere (nw_bn_relu_nhwc_apply_f32): this layer's running ones
        if (num_batches_tracked && threadIdx.x == 0) *num_batches_tracked += 1;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * save_mean[c];
            if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * batch_var[c] * unbias;
        }
    }
    __syncthreads();
    const int q4 = C >> 2;
    const int64_t total = R * q4;
    float mx = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / q4;
        const int c = (int)(idx - r * q4) * 4;
        float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float4 m = *reinterpret_cast<const float4*>(prm + c), a = *reinterpret_cast<const float4*>(prm + C + c),
                     b = *reinterpret_cast<const float4*>(prm + 2 * C + c);
        v.x = __builtin_fmaf(v.x - m.x, a.x, b.x); v.y = __builtin_fmaf(v.y - m.y, a.y, b.y);
        v.z = __builtin_fmaf(v.z - m.z, a.z, b.z); v.w = __builtin_fmaf(v.w - m.w, a.w, b.w);
        if (RELU) {   // (keeps a NaN, like torch's relu)
            v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
        }
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        *reinterpret_cast<float4*>(y + r * C + c) = v;
    }
    mx = block_max(mx, red);
    if (amax && threadIdx.x == 0)
        for (int k = blockIdx.x; k < BN_SLOTS; k += gridDim.x) amax[k] = k == (int)blockIdx.x ? mx : 0.f;
}

// backward partial sums of rows [r0, r1): part[(k * G + g) * C + c], k = 0 sum g, 1 sum g xhat
template <bool RELU>
__global__ __launch_bounds__(1024) void nw_bn_nhwc_bwd_stats_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                                     float* __restrict__ part, int64_t R, int C, int G,
                                                                     int64_t rows_per_chunk) {
    __shared__ float sh[2][BN_RL][BN_TC + 1];
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int c0 = blockIdx.y * BN_TC + 4 * cq;
    const int g = blockIdx.x;
    const int64_t r0 = (int64_t)g * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
        const float4 m4 = *reinterpret_cast<const float4*>(save_mean + c0), i4 = *reinterpret_cast<const float4*>(save_invstd + c0),
                     g4 = *reinterpret_cast<const float4*>(gamma + c0), b4 = *reinterpret_cast<const float4*>(beta + c0);
        const float mean[4] = {m4.x, m4.y, m4.z, m4.w}, inv[4] = {i4.x, i4.y, i4.z, i4.w};
        const float a[4] = {g4.x * i4.x, g4.y * i4.y, g4.z * i4.z, g4.w * i4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
        for (int64_t r = r0 + rl; r < r1; r += BN_RL) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c0);
            const float4 d = *reinterpret_cast<const float4*>(dy + r * C + c0);
            const float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pre = __builtin_fmaf(xv[j] - mean[j], a[j], b[j]);   // the forward's own y: same ReLU mask
                const float gd = (!RELU || pre > 0.f) ? dv[j] : 0.f;
                s1[j] += gd;
                s2[j] = __builtin_fmaf(gd, (xv[j] - mean[j]) * inv[j], s2[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sh[0][rl][4 * cq + j] = s1[j];
        sh[1][rl][4 * cq + j] = s2[j];
    }
    __syncthreads();
    if (tid < BN_TC && blockIdx.y * BN_TC + tid < C) {
        float a1 = sh[0][0][tid], a2 = sh[1][0][tid];
        for (int l = 1; l < BN_RL; ++l) { a1 += sh[0][l][tid]; a2 += sh[1][l][tid]; }
        const int c = blockIdx.y * BN_TC + tid;
        part[((int64_t)0 * G + g) * C + c] = a1;
        part[((int64_t)1 * G + g) * C + c] = a2;
    }
}

// dgamma, dbeta and the two means (k[c], k[C + c]) the apply pass needs: the same 64 x 16 split of the chunk sums
__global__ __launch_bounds__(1024) void nw_bn_nhwc_bwd_finalize_kernel(const float* __restrict__ part, int G, int C, float inv_m,
                                                                        float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                        float* __restrict__ k) {
    __shared__ float sh[2][16][64];
    const int cl = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s1 = 0.f, s2 = 0.f;
    if (c < C)
        for (int g = j; g < G; g += 16) {
            s1 += part[(int64_t)g * C + c];
            s2 += part[((int64_t)G + g) * C + c];
        }
    sh[0][j][cl] = s1; sh[1][j][cl] = s2;
    __syncthreads();
    if (j != 0 || c >= C) return;
    for (int kk = 1; kk < 16; ++kk) { s1 += sh[0][kk][cl]; s2 += sh[1][kk][cl]; }
    dbeta[c] = s1;
    dgamma[c] = s2;
    k[c] = s1 * inv_m;
    k[C + c] = s2 * inv_m;
}

// The backward sums of MANY groups (what a data-gradient convolution's epilogue leaves): dgamma, dbeta and the two means,
// four channels x 1024 group lanes per workgroup, summed in a fixed order.
__global__ __launch_bounds__(1024) void nw_bn_nhwc_bwd_sum_groups_kernel(const float* __restrict__ part, int G, int C, float inv_m,
                                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                          float* __restrict__ k) {
    __shared__ float4 sh[1024];
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * 4;
    auto add4 = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    auto block_sum4 = [&](float4 v) {
        __syncthreads();
        sh[t] = v;
        __syncthreads();
        for (int w = 512; w > 0; w >>= 1) {
            if (t < w) sh[t] = add4(sh[t], sh[t + w]);
            __syncthreads();
        }
        return sh[0];
    };
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (int g = t; g < G; g += 1024) {
        s1 = add4(s1, *reinterpret_cast<const float4*>(part + (int64_t)g * C + c0));
        s2 = add4(s2, *reinterpret_cast<const float4*>(part + ((int64_t)G + g) * C + c0));
    }
    const float4 t1 = block_sum4(s1);
    const float4 t2 = block_sum4(s2);
    if (t != 0) return;
    *reinterpret_cast<float4*>(dbeta + c0) = t1;
    *reinterpret_cast<float4*>(dgamma + c0) = t2;
    *reinterpret_cast<float4*>(k + c0) = make_float4(t1.x * inv_m, t1.y * inv_m, t1.z * inv_m, t1.w * inv_m);
    *reinterpret_cast<float4*>(k + C + c0) = make_float4(t2.x * inv_m, t2.y * inv_m, t2.z * inv_m, t2.w * inv_m);
}

// dx[r][c] = a (g - k1 - xhat k2) [+ acc[r][c]]; amax record of dx
template <bool RELU>
__global__ __launch_bounds__(1024) void nw_bn_nhwc_bwd_apply_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                                     const float* __restrict__ k, const float* acc, int64_t ldacc,
                                                                     float* dx, int64_t lddx, float* __restrict__ amax, int64_t R, int C,
                                                                     const float* __restrict__ part, int G, float inv_m,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    extern __shared__ float prm[];   // [6][C]: mean, invstd, a, beta, k1, k2
    __shared__ float red[16];
    if (G > 0) {
        // few chunks (small maps, C <= 1024): every workgroup sums them itself -- 1024 / C chunk lanes per channel, their
        // partial sums in order -- and workgroup 0 leaves dgamma / dbeta: one launch less per call
        __shared__ float fin[2][1024];
        const int nj = 1024 / C, j = threadIdx.x / C, c = threadIdx.x - j * C;
        float p1 = 0.f, p2 = 0.f;
        if (j < nj)
            for (int g = j; g < G; g += nj) {
                p1 += part[(int64_t)g * C + c];
                p2 += part[((int64_t)G + g) * C + c];
            }
        fin[0][threadIdx.x] = p1; fin[1][threadIdx.x] = p2;
        __syncthreads();
        if ((int)threadIdx.x < C) {
            float s1 = fin[0][c], s2 = fin[1][c];
            for (int jj = 1; jj < nj; ++jj) { s1 += fin[0][jj * C + c]; s2 += fin[1][jj * C + c]; }
            prm[4 * C + c] = s1 * inv_m;
            prm[5 * C + c] = s2 * inv_m;
            if (blockIdx.x == 0) { dbeta[c] = s1; dgamma[c] = s2; }
        }
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float inv = save_invstd[c];
        prm[c] = save_mean[c];
        prm[C + c] = inv;
        prm[2 * C + c] = gamma[c] * inv;
        prm[3 * C + c] = beta[c];
        if (G <= 0) {
            prm[4 * C + c] = k[c];
            prm[5 * C + c] = k[C + c];
        }
    }
    __syncthreads();
    const int q4 = C >> 2;
    const int64_t total = R * q4;
    float mx = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / q4;
        const int c = (int)(idx - r * q4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float4 d = *reinterpret_cast<const float4*>(dy + r * C + c);
        const float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {d.x, d.y, d.z, d.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mean = prm[c + j], inv = prm[C + c + j], a = prm[2 * C + c + j], b = prm[3 * C + c + j];
            const float pre = __builtin_fmaf(xv[j] - mean, a, b);
            const float gd = (!RELU || pre > 0.f) ? dv[j] : 0.f;
            o[j] = a * (gd - prm[4 * C + c + j] - (xv[j] - mean) * inv * prm[5 * C + c + j]);
        }
        if (acc) {
            const float4 e = *reinterpret_cast<const float4*>(acc + r * ldacc + c);
            o[0] += e.x; o[1] += e.y; o[2] += e.z; o[3] += e.w;
        }
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        *reinterpret_cast<float4*>(dx + r * lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    mx = block_max(mx, red);
    if (amax && threadIdx.x == 0)
        for (int kk = blockIdx.x; kk < BN_SLOTS; kk += gridDim.x) amax[kk] = kk == (int)blockIdx.x ? mx : 0.f;
}

inline void stats_grid(int64_t R, int64_t C, int* G, int64_t* rpc) {
    int64_t g = BN_GC / C;                        // chunks x channels bounded: the apply passes re-read all of them
    if (g > 256) g = 256;
    const int64_t gmax = (R + 4 * BN_RL - 1) / (4 * BN_RL);   // at least 4 rows per row lane
    if (g > gmax) g = gmax;
    if (g < 1) g = 1;
    *rpc = (R + g - 1) / g;
    *G = (int)((R + *rpc - 1) / *rpc);
}
inline int apply_grid(int64_t R, int64_t C) {
    const int64_t want = (R * (C / 4) + 1023) / 1024;
    return (int)(want < 1 ? 1 : (want > BN_SLOTS ? BN_SLOTS : want));
}
inline bool bad_align(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
             reinterpret_cast<uintptr_t>(d)) & 15) != 0;
}

}  // namespace
}  // namespace nw

namespace nw {
// bn_dgrad.hip: dgamma, dbeta and the two means from G groups of (sum g, sum g xhat) rows -- part[(k * G + g) * C + c]
int bn_bwd_finalize_groups(const float* part, int G, int C, float inv_m, float* dgamma, float* dbeta, float* k, hipStream_t st) {
    if (!part || G <= 0 || C <= 0 || !dgamma || !dbeta || !k) return NW_ERR_INVALID_ARG;
    hipLaunchKernelGGL(nw_bn_nhwc_bwd_finalize_kernel, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, st, part, G, C, inv_m, dgamma,
                       dbeta, k);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
}  // namespace nw

extern "C" size_t nw_bn_nhwc_workspace_bytes(int64_t rows, int64_t c) {
    if (rows <= 0 || c <= 0) return 0;
    int G; int64_t rpc;
    nw::stats_grid(rows, c, &G, &rpc);
    return ((size_t)3 * G * c + 2 * c) * sizeof(float);
}

extern "C" int nw_bn_relu_nhwc_train_fwd_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* running_mean,
                                             float* running_var, float* y, float* save_mean, float* save_invstd,
                                             int64_t* num_batches_tracked, float* amax_out, void* workspace, size_t workspace_bytes,
                                             int64_t rows, int64_t c, float momentum, float eps, int relu, void* stream) {
    using namespace nw;
    if (rows < 0 || c <= 0 || c % 4 || ldx < c || ldx % 4) return NW_ERR_INVALID_ARG;
    if (rows == 0) return NW_OK;
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd) return NW_ERR_INVALID_ARG;
    if (bad_align(x, y, amax_out, workspace) || bad_align(gamma, beta, save_mean, save_invstd)) return NW_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < nw_bn_nhwc_workspace_bytes(rows, c)) return NW_ERR_WORKSPACE;
    if (c > BN_MAX_C) return NW_ERR_UNSUPPORTED;   // the per-channel factors live in LDS
    hipStream_t st = static_cast<hipStream_t>(stream);
    int G; int64_t rpc;
    stats_grid(rows, c, &G, &rpc);
    float* part = static_cast<float*>(workspace);
    const unsigned ct = (unsigned)((c + BN_TC - 1) / BN_TC);
    hipLaunchKernelGGL(nw_bn_nhwc_stats_kernel, dim3((unsigned)G, ct), dim3(1024), 0, st, x, ldx, part, rows, (int)c, G, rpc, 0);
    hipLaunchKernelGGL(nw_bn_nhwc_finalize_kernel, dim3((unsigned)((c + 63) / 64)), dim3(1024), 0, st, part, G, (int)c, running_mean,
                       running_var, save_mean, save_invstd, num_batches_tracked, momentum, eps, (float*)nullptr);
    const int ag = apply_grid(rows, c);
    const size_t lds = (size_t)3 * c * sizeof(float);
    if (relu)
        hipLaunchKernelGGL((nw_bn_nhwc_apply_kernel<true>), dim3((unsigned)ag), dim3(1024), lds, st, x, ldx, gamma, beta, save_mean,
                           save_invstd, y, amax_out, rows, (int)c, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (int64_t*)nullptr, 0.f, 1.f);
    else
        hipLaunchKernelGGL((nw_bn_nhwc_apply_kernel<false>), dim3((unsigned)ag), dim3(1024), lds, st, x, ldx, gamma, beta, save_mean,
                           save_invstd, y, amax_out, rows, (int)c, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (int64_t*)nullptr, 0.f, 1.f);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

// ---- the forward in phases, for callers that get the batch statistics elsewhere (a dense block: the statistics of a
// channel do not change from layer to layer, and a convolution's epilogue leaves the moments of what it writes)
extern "C" int nw_bn_nhwc_moments_f32(const float* x, int64_t ldx, int64_t rows, int64_t c, float eps, float* mean, float* invstd,
                                      float* var, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace nw;
    if (rows <= 0 || c <= 0 || c % 4 || ldx < c || ldx % 4) return NW_ERR_INVALID_ARG;
    if (!x || !mean || !invstd || !var) return NW_ERR_INVALID_ARG;
    if (bad_align(x, workspace) || bad_align(mean, invstd, var)) return NW_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < nw_bn_nhwc_workspace_bytes(rows, c)) return NW_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int G; int64_t rpc;
    stats_grid(rows, c, &G, &rpc);
    float* part = static_cast<float*>(workspace);
    const unsigned ct = (unsigned)((c + BN_TC - 1) / BN_TC);
    hipLaunchKernelGGL(nw_bn_nhwc_stats_kernel, dim3((unsigned)G, ct), dim3(1024), 0, st, x, ldx, part, rows, (int)c, G, rpc, 0);
    hipLaunchKernelGGL(nw_bn_nhwc_finalize_kernel, dim3((unsigned)((c + 63) / 64)), dim3(1024), 0, st, part, G, (int)c, (float*)nullptr,
                       (float*)nullptr, mean, invstd, (int64_t*)nullptr, 0.f, eps, var);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_bn_nhwc_moments_from_partials_f32(float* partials, int64_t groups, int64_t c, float eps, float* mean,
                                                    float* invstd, float* var, void* stream) {
    using namespace nw;
    if (groups <= 0 || groups >= (1LL << 30) || c <= 0) return NW_ERR_INVALID_ARG;
    if (!partials || !mean || !invstd || !var) return NW_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c % 4 || (reinterpret_cast<uintptr_t>(partials) & 15)) return NW_ERR_INVALID_ARG;
    hipLaunchKernelGGL(nw_bn_nhwc_merge_groups_kernel, dim3((unsigned)(c / 4)), dim3(1024), 0, st, partials, (int)groups, (int)c, eps,
                       mean, invstd, var, (float*)nullptr, (float*)nullptr, BnPrep{});
    NW_CHECK_LAUNCH();
    return NW_OK;
}

// ---- round 4: BatchNorm + ReLU applied by the consuming convolution's loaders (nw_conv2d_nhwc_bnrelu_f16x2).  These entries
// produce what it takes: the per-channel table mean | a | beta (3 c floats) and the amax record bounding |relu(bn(x))|, from
// statistics that include each channel's minimum and maximum.
static bool prep_args_ok(const float* gamma, const float* beta, float* tab, float* amax) {
    return gamma && beta && tab && amax && !nw::bad_align(tab, amax);
}
extern "C" int nw_bn_nhwc_moments_minmax_f32(const float* x, int64_t ldx, int64_t rows, int64_t c, float eps, float* mean,
                                             float* invstd, float* var, float* vmin, float* vmax, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    using namespace nw;
    if (rows <= 0 || c <= 0 || c % 4 || ldx < c || ldx % 4) return NW_ERR_INVALID_ARG;
    if (!x || !mean || !invstd || !var || !vmin || !vmax) return NW_ERR_INVALID_ARG;
    if (bad_align(x, workspace) || bad_align(mean, invstd, var)) return NW_ERR_INVALID_ARG;
    int G; int64_t rpc;
    stats_grid(rows, c, &G, &rpc);
    if (!workspace || workspace_bytes < ((size_t)5 * G * c + 2 * c) * sizeof(float)) return NW_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    const unsigned ct = (unsigned)((c + BN_TC - 1) / BN_TC);
    hipLaunchKernelGGL(nw_bn_nhwc_stats_kernel, dim3((unsigned)G, ct), dim3(1024), 0, st, x, ldx, part, rows, (int)c, G, rpc, 1);
    hipLaunchKernelGGL(nw_bn_nhwc_finalize_kernel, dim3((unsigned)((c + 63) / 64)), dim3(1024), 0, st, part, G, (int)c, (float*)nullptr,
                       (float*)nullptr, mean, invstd, (int64_t*)nullptr, 0.f, eps, var, vmin, vmax);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
extern "C" size_t nw_bn_nhwc_minmax_workspace_bytes(int64_t rows, int64_t c) {
    if (rows <= 0 || c <= 0) return 0;
    int G; int64_t rpc;
    nw::stats_grid(rows, c, &G, &rpc);
    return ((size_t)5 * G * c + 2 * c) * sizeof(float);
}

/* partials: 5 rows per group (count, mean, M2, minimum, maximum: what nw_conv2d_nhwc_f16x2 / _bnrelu_f16x2 leave in `moments`).
 * gamma == NULL: statistics only (mean, invstd, var, vmin, vmax); else also the table and bound of the BatchNorm (gamma, beta)
 * that reads this tensor next, and that layer's running statistics (nullable) and step counter (nullable). */
extern "C" int nw_bn_nhwc_prep_window_from_partials_f32(float* partials, int64_t groups, int64_t c, int64_t offset, int64_t n_old,
                                                        int64_t rows, float eps, float* mean, float* invstd, float* var, float* vmin,
                                                        float* vmax, const float* gamma, const float* beta, float* running_mean,
                                                        float* running_var, int64_t* num_batches_tracked, float momentum, int relu,
                                                        float* tab, float* amax_out, void* stream) {
    using namespace nw;
    if (groups <= 0 || groups >= (1LL << 30) || c <= 0 || c % 4 || offset < 0 || n_old < 0 || n_old > offset) return NW_ERR_INVALID_ARG;
    if (!partials || !mean || !invstd || !var || !vmin || !vmax || (reinterpret_cast<uintptr_t>(partials) & 15)) return NW_ERR_INVALID_ARG;
    MergeP p{};
    p.part = partials; p.G = (int)groups; p.C = (int)c; p.off = (int)offset; p.eps = eps;
    p.mean = mean; p.invstd = invstd; p.var = var; p.vmin = vmin; p.vmax = vmax;
    p.nold = 0; p.tc = (int)(offset + c);
    p.unbias_old = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
    int64_t grid = (c + 15) / 16;
    if (gamma) {
        if (!prep_args_ok(gamma, beta, tab, amax_out)) return NW_ERR_INVALID_ARG;
        p.prep = BnPrep{gamma, beta, tab, amax_out, running_mean, running_var, num_batches_tracked, momentum, relu};
        p.nold = (int)n_old;
        grid += (n_old + 1023) / 1024;
        if (grid > BN_SLOTS) return NW_ERR_UNSUPPORTED;         // (one amax slot per workgroup)
    }
    hipLaunchKernelGGL(nw_bn_nhwc_merge_prep_kernel, dim3((unsigned)grid), dim3(1024), 0, static_cast<hipStream_t>(stream), p);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
extern "C" int nw_bn_nhwc_prep_from_partials_f32(float* partials, int64_t groups, int64_t c, float eps, float* mean, float* invstd,
                                                 float* var, float* vmin, float* vmax, const float* gamma, const float* beta,
                                                 float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                                 float momentum, int relu, float* tab, float* amax_out, void* stream) {
    return nw_bn_nhwc_prep_window_from_partials_f32(partials, groups, c, 0, 0, 2, eps, mean, invstd, var, vmin, vmax, gamma, beta,
                                                    running_mean, running_var, num_batches_tracked, momentum, relu, tab, amax_out, stream);
}

/* the table and bound of a BatchNorm (gamma, beta) over channels whose statistics exist (rows = samples per channel, for the
 * unbiased running variance) */
extern "C" int nw_bn_nhwc_prep_f32(const float* mean, const float* invstd, const float* var, const float* vmin, const float* vmax,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float momentum, int relu, int64_t rows, int64_t c, float* tab,
                                   float* amax_out, void* stream) {
    using namespace nw;
    if (c <= 0 || rows <= 0 || !mean || !invstd || !var || !vmin || !vmax || !prep_args_ok(gamma, beta, tab, amax_out)) return NW_ERR_INVALID_ARG;
    if ((c + 255) / 256 > BN_SLOTS) return NW_ERR_UNSUPPORTED;
    const BnPrep pr{gamma, beta, tab, amax_out, running_mean, running_var, num_batches_tracked, momentum, relu};
    const float unbias = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
    hipLaunchKernelGGL(nw_bn_nhwc_prep_kernel, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), mean,
                       invstd, var, vmin, vmax, (int)c, unbias, pr);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_bn_relu_nhwc_apply_f32(const float* x, int64_t ldx, const float* mean, const float* invstd, const float* var,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                                         int64_t* num_batches_tracked, float momentum, float* y, float* amax_out, int64_t rows,
                                         int64_t c, int relu, void* stream) {
    using namespace nw;
    if (rows <= 0 || c <= 0 || c % 4 || ldx < c || ldx % 4) return NW_ERR_INVALID_ARG;
    if (!x || !mean || !invstd || (!var && running_var) || !gamma || !beta || !y) return NW_ERR_INVALID_ARG;   // (var: only for the running variance)
    if (bad_align(x, y, amax_out) || bad_align(gamma, beta, mean, invstd)) return NW_ERR_INVALID_ARG;
    if (c > BN_MAX_C) return NW_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ag = apply_grid(rows, c);
    const size_t lds = (size_t)3 * c * sizeof(float);
    const float unbias = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
    if (relu)
        hipLaunchKernelGGL((nw_bn_nhwc_apply_kernel<true>), dim3((unsigned)ag), dim3(1024), lds, st, x, ldx, gamma, beta, mean, invstd, y,
                           amax_out, rows, (int)c, var, running_mean, running_var, num_batches_tracked, momentum, unbias);
    else
        hipLaunchKernelGGL((nw_bn_nhwc_apply_kernel<false>), dim3((unsigned)ag), dim3(1024), lds, st, x, ldx, gamma, beta, mean, invstd, y,
                           amax_out, rows, (int)c, var, running_mean, running_var, num_batches_tracked, momentum, unbias);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_bn_relu_nhwc_train_bwd_f32(const float* x, int64_t ldx, const float* dy, const float* gamma, const float* beta,
                                             const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                                             float* dbeta, const float* acc, int64_t ldacc, int64_t lddx, float* amax_out,
                                             void* workspace, size_t workspace_bytes, int64_t rows, int64_t c, int relu,
                                             void* stream) {
    using namespace nw;
    if (lddx == 0) lddx = c;
    if (rows < 0 || c <= 0 || c % 4 || ldx < c || ldx % 4 || (acc && (ldacc < c || ldacc % 4)) || lddx < c || lddx % 4)
        return NW_ERR_INVALID_ARG;
    if (rows == 0) return NW_OK;
    if (!x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta) return NW_ERR_INVALID_ARG;
    if (bad_align(x, dy, dx, acc) || bad_align(gamma, beta, save_mean, save_invstd) || bad_align(amax_out, workspace))
        return NW_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < nw_bn_nhwc_workspace_bytes(rows, c)) return NW_ERR_WORKSPACE;
    if (c > BN_MAX_C) return NW_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int G; int64_t rpc;
    stats_grid(rows, c, &G, &rpc);
    // small maps: at most 8192 chunk sums, merged by the apply workgroups themselves (no finalize launch)
    const bool inline_fin = rows <= BN_INLINE_ROWS && c <= 1024 && knob(KNOB_BN_INLINE_FIN) != 0;
    if (inline_fin && (int64_t)G * c > BN_INLINE_GC) {
        int64_t g = BN_INLINE_GC / c;
        if (g < 1) g = 1;
        rpc = (rows + g - 1) / g;
        G = (int)((rows + rpc - 1) / rpc);
    }
    float* part = static_cast<float*>(workspace);
    const unsigned ct = (unsigned)((c + BN_TC - 1) / BN_TC);
    const int ag = apply_grid(rows, c);
    const size_t lds = (size_t)6 * c * sizeof(float);
    float* k = part + (size_t)3 * G * c;
#define NW_BNB(R_)                                                                                                             \
    do {                                                                                                                       \
        hipLaunchKernelGGL((nw_bn_nhwc_bwd_stats_kernel<R_>), dim3((unsigned)G, ct), dim3(1024), 0, st, x, ldx, dy, gamma, beta, \
                           save_mean, save_invstd, part, rows, (int)c, G, rpc);                                               \
        if (!inline_fin)                                                                                                       \
            hipLaunchKernelGGL(nw_bn_nhwc_bwd_finalize_kernel, dim3((unsigned)((c + 63) / 64)), dim3(1024), 0, st, part, G,    \
                               (int)c, 1.f / (float)rows, dgamma, dbeta, k);                                                   \
        hipLaunchKernelGGL((nw_bn_nhwc_bwd_apply_kernel<R_>), dim3((unsigned)ag), dim3(1024), lds, st, x, ldx, dy, gamma, beta,  \
                           save_mean, save_invstd, k, acc, ldacc, dx, lddx, amax_out, rows, (int)c, part, inline_fin ? G : 0,  \
                           1.f / (float)rows, dgamma, dbeta);                                                                  \
    } while (0)
    if (relu) NW_BNB(true); else NW_BNB(false);
#undef NW_BNB
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_bn_relu_nhwc_train_bwd_from_partials_f32(const float* x, int64_t ldx, const float* dy, const float* gamma,
                                                           const float* beta, const float* save_mean, const float* save_invstd,
                                                           const float* partials, int64_t groups, float* dx, float* dgamma,
                                                           float* dbeta, const float* acc, int64_t ldacc, int64_t lddx,
                                                           float* amax_out, void* workspace, size_t workspace_bytes, int64_t rows,
                                                           int64_t c, void* stream) {
    using namespace nw;
    if (lddx == 0) lddx = c;
    if (rows <= 0 || c <= 0 || c % 4 || ldx < c || ldx % 4 || (acc && (ldacc < c || ldacc % 4)) || lddx < c || lddx % 4 ||
        groups <= 0 || groups >= (1LL << 30))
        return NW_ERR_INVALID_ARG;
    if (!x || !dy || !gamma || !beta || !save_mean || !save_invstd || !partials || !dx || !dgamma || !dbeta) return NW_ERR_INVALID_ARG;
    if (bad_align(x, dy, dx, acc) || bad_align(gamma, beta, save_mean, save_invstd) || bad_align(amax_out, workspace, partials) ||
        bad_align(dgamma, dbeta))
        return NW_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < (size_t)2 * c * sizeof(float)) return NW_ERR_WORKSPACE;
    if (c > BN_MAX_C) return NW_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* k = static_cast<float*>(workspace);
    hipLaunchKernelGGL(nw_bn_nhwc_bwd_sum_groups_kernel, dim3((unsigned)(c / 4)), dim3(1024), 0, st, partials, (int)groups, (int)c,
                       1.f / (float)rows, dgamma, dbeta, k);
    const int ag = apply_grid(rows, c);
    const size_t lds = (size_t)6 * c * sizeof(float);
    hipLaunchKernelGGL((nw_bn_nhwc_bwd_apply_kernel<true>), dim3((unsigned)ag), dim3(1024), lds, st, x, ldx, dy, gamma, beta, save_mean,
                       save_invstd, k, acc, ldacc, dx, lddx, amax_out, rows, (int)c, (const float*)nullptr, 0, 0.f, (float*)nullptr,
                       (float*)nullptr);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
