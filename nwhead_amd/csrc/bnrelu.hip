// bnrelu.hip -- eval-mode BatchNorm (+ ReLU) of the pre-activation backbones as ONE pass
// (gfx950 / MI355X only).
//
// DenseNet applies BatchNorm -> ReLU -> conv (reference model/densenet.py:33-60, :82-91, :139): the
// BatchNorm cannot be folded into a convolution, but at inference it is y = max(a_c x + b_c, 0) with
// a_c = gamma / sqrt(var + eps), b_c = beta - mean a_c; torch runs it as two kernels (batch norm, then
// relu), each a full read + write of the activation.  The input may be the first c channels of a wider
// slab (the concat-free dense block): rows (n, c) are hw contiguous floats, batch stride given.
#include "nw_internal.h"
// max(v, 0) that keeps a NaN, like torch's relu (fmaxf(NaN, 0) is 0)
#define NW_RELU(v) ((v) < 0.f ? 0.f : (v))

namespace nw {
namespace {

template <bool RELU, bool VEC>
__global__ __launch_bounds__(256) void nw_scale_shift_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              float* __restrict__ out, int64_t total, int64_t C,
                                                              int64_t hw, int64_t x_batch_stride) {
    // one item = 4 floats (VEC) or 1 float of an (n, c) plane
    const int64_t per = VEC ? hw / 4 : hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t plane = i / per, within = i - plane * per;
        const int64_t n = plane / C, c = plane - n * C;
        const float a = scale[c], b = shift[c];
        if (VEC) {
            float4 v = *reinterpret_cast<const float4*>(x + n * x_batch_stride + c * hw + 4 * within);
            v.x = __builtin_fmaf(v.x, a, b); v.y = __builtin_fmaf(v.y, a, b);
            v.z = __builtin_fmaf(v.z, a, b); v.w = __builtin_fmaf(v.w, a, b);
            if (RELU) { v.x = NW_RELU(v.x); v.y = NW_RELU(v.y); v.z = NW_RELU(v.z); v.w = NW_RELU(v.w); }
            *reinterpret_cast<float4*>(out + plane * hw + 4 * within) = v;
        } else {
            float v = __builtin_fmaf(x[n * x_batch_stride + c * hw + within], a, b);
            if (RELU) v = NW_RELU(v);
            out[plane * hw + within] = v;
        }
    }
}

// max(a_c x + b_c, 0) followed by a 2x2 / stride-2 average pool, one output pixel per thread (two 8-byte reads).
// In the transitions of the folded DenseNets (norm - relu - conv 1x1 - avgpool, model/densenet.py:82-91,
// model/densenet3.py:25-35) the pool commutes with the bias-free 1x1 convolution, so it runs FIRST and the
// convolution sees a quarter of the pixels.  Output (n, C, h/2, w/2) contiguous (floor, like torch's avg_pool2d).
__global__ __launch_bounds__(256) void nw_scale_shift_relu_pool2_kernel(const float* __restrict__ x,
                                                                         const float* __restrict__ scale,
                                                                         const float* __restrict__ shift,
                                                                         float* __restrict__ out, int64_t total, int C,
                                                                         int h, int w, int64_t x_batch_stride, int relu) {
    const int ho = h / 2, wo = w / 2;
    const float lo = relu ? 0.f : -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xo = (int)(i % wo);
        const int64_t t = i / wo;
        const int yo = (int)(t % ho);
        const int64_t plane = t / ho;
        const int64_t n = plane / C;
        const int c = (int)(plane - n * C);
        const float a = scale[c], b = shift[c];
        const float* p = x + n * x_batch_stride + ((int64_t)c * h + 2 * yo) * w + 2 * xo;
        auto clamp = [lo](float v) { return v < lo ? lo : v; };   // keeps a NaN (fmaxf would drop it)
        const float v0 = clamp(__builtin_fmaf(p[0], a, b)), v1 = clamp(__builtin_fmaf(p[1], a, b));
        const float v2 = clamp(__builtin_fmaf(p[w], a, b)), v3 = clamp(__builtin_fmaf(p[w + 1], a, b));
        out[i] = ((v0 + v1) + (v2 + v3)) * 0.25f;
    }
}


// ---------------------------------------------------------------------------------------------------
// Training-mode BatchNorm2d (+ ReLU), forward and backward, one workgroup per channel.
// torch runs the pair as batch-norm kernels (MIOpen: mean/variance, final, normalise; backward: three more)
// plus a relu kernel each way: measured 9.2 ms of a 27 ms DenseNet-121 step (121 pairs).  Here the forward is
// one kernel (shifted single-pass sums, then normalise + ReLU on the second read, which comes from L2 for all
// but the stem) and the backward one kernel (sums of g and g*xhat, then dx); the ReLU mask is recomputed from
// x with the forward's own arithmetic, so no activation is saved for it.
// Layout: x element (i, c, p) at x[i*x_batch_stride + c*hw + p]; y, dy, dx are (n, C, hw) contiguous.
template <bool VEC, typename F>
__device__ __forceinline__ void for_channel(const float* __restrict__ base, int64_t n, int64_t hw, int64_t bstride,
                                            int tid, int nthr, F&& f) {
    // f(plane index i, offset inside the plane, value(s)) over the n planes of one channel.  32-bit index
    // arithmetic (a channel has < 2^31 items: checked by the launchers) and four items in flight per thread.
    const unsigned q = (unsigned)(VEC ? hw / 4 : hw), total = (unsigned)n * q;
#pragma unroll 4
    for (unsigned idx = tid; idx < total; idx += nthr) {
        const unsigned i = idx / q, j = (idx - i * q) * (VEC ? 4u : 1u);
        const float* p = base + (int64_t)i * bstride + j;
        if (VEC) {
            f((int64_t)i, (int64_t)j, *reinterpret_cast<const float4*>(p));
        } else {
            f((int64_t)i, (int64_t)j, make_float4(*p, 0.f, 0.f, 0.f));
        }
    }
}

// RES: y = relu(bn(x) + residual) -- the tail of a ResNet block (model/resnet.py:58-66, :100-108); residual and its
// gradient are (n, C, hw) contiguous like y.
template <bool RELU, bool VEC, bool RES>
__global__ __launch_bounds__(1024) void nw_bn_train_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ residual, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ y,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, int64_t* __restrict__ num_batches_tracked,
    int64_t n, int64_t C, int64_t hw, int64_t x_batch_stride, float momentum, float eps) {
    __shared__ float red[16];
    const int64_t c = blockIdx.x;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (num_batches_tracked && c == 0 && tid == 0) *num_batches_tracked += 1;  // BatchNorm2d's step counter, no launch of its own
    const float* xc = x + c * hw;
    // One pass of SHIFTED sums, S1 = sum(x - K), S2 = sum((x - K)^2), var = S2/m - (S1/m)^2.  Unshifted
    // (K = 0) this loses the variance of a channel whose spread is small next to its offset, and
    // 1/sqrt(var + 1e-5) amplifies that; with K within a few standard deviations of the mean the
    // cancellation is harmless ((mean - K)^2 / var enters the relative error once, times 2^-24).  K is the
    // mean of 64 samples spread over the channel's planes (a single sample could be a border value).
    __shared__ float k_s;
    if (tid < 64) {
        const int64_t i = tid % n, j = ((int64_t)tid * 37) % hw;
        const float ks = wave_sum(xc[i * x_batch_stride + j]);
        if (tid == 0) k_s = ks * (1.f / 64.f);
    }
    __syncthreads();
    const float K = k_s;
    float s1 = 0.f, s2 = 0.f;
    for_channel<VEC>(xc, n, hw, x_batch_stride, tid, nthr, [&](int64_t, int64_t, const float4 v) {
        const float d0 = v.x - K;
        s1 += d0; s2 = __builtin_fmaf(d0, d0, s2);
        if (VEC) {
            const float d1 = v.y - K, d2 = v.z - K, d3 = v.w - K;
            s1 += d1; s2 = __builtin_fmaf(d1, d1, s2);
            s1 += d2; s2 = __builtin_fmaf(d2, d2, s2);
            s1 += d3; s2 = __builtin_fmaf(d3, d3, s2);
        }
    });
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float m = (float)(n * hw);
    const float md = s1 / m;
    const float mean = K + md;
    const float var = fmaxf(s2 / m - md * md, 0.f);  // biased (normalisation); the running one is unbiased
    const float invstd = 1.f / sqrtf(var + eps);
    if (tid == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (m > 1.f ? m / (m - 1.f) : 1.f);
    }
    // y = (x - mean) * a + beta, not x * a + (beta - mean * a): the latter rounds at the size of mean * a, which a
    // near-constant channel (a ~ 1/sqrt(eps)) turns into visible error
    const float a = gamma[c] * invstd, b = beta[c];
    float* yc = y + c * hw;
    const float* rc = RES ? residual + c * hw : nullptr;
    for_channel<VEC>(xc, n, hw, x_batch_stride, tid, nthr, [&](int64_t i, int64_t j, float4 v) {
        v.x = __builtin_fmaf(v.x - mean, a, b);
        if (VEC) {
            v.y = __builtin_fmaf(v.y - mean, a, b); v.z = __builtin_fmaf(v.z - mean, a, b); v.w = __builtin_fmaf(v.w - mean, a, b);
            if (RES) {
                const float4 r = *reinterpret_cast<const float4*>(rc + i * C * hw + j);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (RELU) { v.x = NW_RELU(v.x); v.y = NW_RELU(v.y); v.z = NW_RELU(v.z); v.w = NW_RELU(v.w); }
            *reinterpret_cast<float4*>(yc + i * C * hw + j) = v;
        } else {
            if (RES) v.x += rc[i * C * hw + j];
            if (RELU) v.x = NW_RELU(v.x);
            yc[i * C * hw + j] = v.x;
        }
    });
}

template <bool RELU, bool VEC, bool RES>
__global__ __launch_bounds__(1024) void nw_bn_train_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ residual, const float* __restrict__ dy,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, float* __restrict__ dx, float* __restrict__ dresidual,
    float* __restrict__ dgamma, float* __restrict__ dbeta, const float* __restrict__ acc, int64_t acc_batch_stride,
    int64_t n, int64_t C, int64_t hw, int64_t x_batch_stride) {
    __shared__ float red[16];
    const int64_t c = blockIdx.x;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const float* xc = x + c * hw;
    const float* dyc = dy + c * hw;
    // acc (nullable): another gradient of x, added into dx here instead of by a separate strided add (the running
    // concatenation of a dense block feeds this BatchNorm AND the next concatenation); element (i, c, p) at
    // acc[i * acc_batch_stride + c * hw + p]
    const float* accc = acc ? acc + c * hw : nullptr;
    const float* rc = RES ? residual + c * hw : nullptr;
    float* drc = RES ? dresidual + c * hw : nullptr;
    const float mean = save_mean[c], invstd = save_invstd[c], g = gamma[c];
    const float a = g * invstd, b = beta[c];  // the forward's own y = fma(x - mean, a, beta): same ReLU mask
    float s1 = 0.f, s2 = 0.f;
    auto term = [&](float xv, float rv, float dv, float& gd, float& xh) {
        xh = (xv - mean) * invstd;
        float pre = __builtin_fmaf(xv - mean, a, b);
        if (RES) pre += rv;
        gd = (!RELU || pre > 0.f) ? dv : 0.f;
    };
    for_channel<VEC>(xc, n, hw, x_batch_stride, tid, nthr, [&](int64_t i, int64_t j, const float4 v) {
        float gd, xh;
        if (VEC) {
            const float4 d = *reinterpret_cast<const float4*>(dyc + i * C * hw + j);
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (RES) r = *reinterpret_cast<const float4*>(rc + i * C * hw + j);
            term(v.x, r.x, d.x, gd, xh); s1 += gd; s2 = __builtin_fmaf(gd, xh, s2);
            term(v.y, r.y, d.y, gd, xh); s1 += gd; s2 = __builtin_fmaf(gd, xh, s2);
            term(v.z, r.z, d.z, gd, xh); s1 += gd; s2 = __builtin_fmaf(gd, xh, s2);
            term(v.w, r.w, d.w, gd, xh); s1 += gd; s2 = __builtin_fmaf(gd, xh, s2);
        } else {
            term(v.x, RES ? rc[i * C * hw + j] : 0.f, dyc[i * C * hw + j], gd, xh); s1 += gd; s2 = __builtin_fmaf(gd, xh, s2);
        }
    });
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (tid == 0) {
        dbeta[c] = s1;
        dgamma[c] = s2;
    }
    const float m = (float)(n * hw);
    const float k1 = s1 / m, k2 = s2 / m;
    float* dxc = dx + c * hw;
    for_channel<VEC>(xc, n, hw, x_batch_stride, tid, nthr, [&](int64_t i, int64_t j, const float4 v) {
        float gd, xh;
        if (VEC) {
            const float4 d = *reinterpret_cast<const float4*>(dyc + i * C * hw + j);
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f), o, go;
            if (RES) r = *reinterpret_cast<const float4*>(rc + i * C * hw + j);
            term(v.x, r.x, d.x, gd, xh); o.x = a * (gd - k1 - xh * k2); go.x = gd;
            term(v.y, r.y, d.y, gd, xh); o.y = a * (gd - k1 - xh * k2); go.y = gd;
            term(v.z, r.z, d.z, gd, xh); o.z = a * (gd - k1 - xh * k2); go.z = gd;
            term(v.w, r.w, d.w, gd, xh); o.w = a * (gd - k1 - xh * k2); go.w = gd;
            if (accc) {
                const float4 e = *reinterpret_cast<const float4*>(accc + i * acc_batch_stride + j);
                o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
            }
            *reinterpret_cast<float4*>(dxc + i * C * hw + j) = o;
            if (RES) *reinterpret_cast<float4*>(drc + i * C * hw + j) = go;
        } else {
            term(v.x, RES ? rc[i * C * hw + j] : 0.f, dyc[i * C * hw + j], gd, xh);
            dxc[i * C * hw + j] = a * (gd - k1 - xh * k2) + (accc ? accc[i * acc_batch_stride + j] : 0.f);
            if (RES) drc[i * C * hw + j] = gd;
        }
    });
}

inline unsigned channel_threads(int64_t per_channel) { return per_channel >= 16384 ? 1024u : per_channel >= 2048 ? 512u : 256u; }

// out[r][c] = act(x[r][c] + bias[c] [+ res[r][c]]) for a channels-last activation seen as (rows = n h w, C): what follows
// a bias-free convolution in the folded channels_last ResNets (bias add, identity add and ReLU in ONE pass instead of
// three element-wise launches).  float4 along C (C % 4 == 0); in place allowed.
template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void nw_bias_act_rows_kernel(const float* x, const float* __restrict__ bias,
                                                               const float* __restrict__ res, float* out,   // out may be x (in place)
                                                               int64_t total4, int c4) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % c4);
        float4 v = reinterpret_cast<const float4*>(x)[idx];
        const float4 b = reinterpret_cast<const float4*>(bias)[c];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (RES) {
            const float4 r = reinterpret_cast<const float4*>(res)[idx];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (RELU) {
            v.x = NW_RELU(v.x); v.y = NW_RELU(v.y); v.z = NW_RELU(v.z); v.w = NW_RELU(v.w);
        }
        reinterpret_cast<float4*>(out)[idx] = v;
    }
}

}  // namespace
}  // namespace nw

extern "C" int nw_scale_shift_relu_f32(const float* x, const float* scale, const float* shift, float* out,
                                       int64_t n, int64_t c, int64_t hw, int64_t x_batch_stride, int relu,
                                       void* stream) {
    using namespace nw;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n < 0 || c < 0 || hw < 0 || x_batch_stride < c * hw) return NW_ERR_INVALID_ARG;
    if (n == 0 || c == 0 || hw == 0) return NW_OK;
    if (!x || !scale || !shift || !out) return NW_ERR_INVALID_ARG;
    const bool vec = hw % 4 == 0 && x_batch_stride % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const int64_t total = n * c * (vec ? hw / 4 : hw);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
#define NW_SS(R_, V_)                                                                                              \
    hipLaunchKernelGGL((nw_scale_shift_kernel<R_, V_>), dim3((unsigned)blocks), dim3(256), 0, st, x, scale, shift, \
                       out, total, c, hw, x_batch_stride)
    if (relu) { if (vec) NW_SS(true, true); else NW_SS(true, false); }
    else { if (vec) NW_SS(false, true); else NW_SS(false, false); }
#undef NW_SS
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_scale_shift_relu_avgpool2_f32(const float* x, const float* scale, const float* shift, float* out,
                                                int64_t n, int64_t c, int64_t h, int64_t w, int64_t x_batch_stride,
                                                int relu, void* stream) {
    using namespace nw;
    if (n < 0 || c < 0 || h < 0 || w < 0 || x_batch_stride < c * h * w || c > 0x7fffffffLL || h > 0x7fffffffLL || w > 0x7fffffffLL)
        return NW_ERR_INVALID_ARG;
    const int64_t total = n * c * (h / 2) * (w / 2);
    if (total == 0) return NW_OK;
    if (!x || !scale || !shift || !out) return NW_ERR_INVALID_ARG;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(nw_scale_shift_relu_pool2_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       scale, shift, out, total, (int)c, (int)h, (int)w, x_batch_stride, relu);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_bn_relu_train_fwd_f32(const float* x, const float* residual, const float* gamma, const float* beta, float* running_mean,
                                        float* running_var, float* y, float* save_mean, float* save_invstd,
                                        int64_t* num_batches_tracked, int64_t n, int64_t c, int64_t hw,
                                        int64_t x_batch_stride, float momentum, float eps, int relu, void* stream) {
    using namespace nw;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n <= 0 || c < 0 || hw <= 0 || x_batch_stride < c * hw || c > 0x7fffffffLL || n * hw > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    if (c == 0) return NW_OK;
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd) return NW_ERR_INVALID_ARG;
    const bool vec = hw % 4 == 0 && x_batch_stride % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
    const unsigned thr = channel_threads(n * hw);
    if (residual && !relu) return NW_ERR_UNSUPPORTED;  // only the relu(bn(x) + r) tail exists in the backbones
    if (residual) {
        if (vec)
            hipLaunchKernelGGL((nw_bn_train_fwd_kernel<true, true, true>), dim3((unsigned)c), dim3(thr), 0, st, x, residual, gamma, beta,
                               running_mean, running_var, y, save_mean, save_invstd, num_batches_tracked, n, c, hw, x_batch_stride, momentum, eps);
        else
            hipLaunchKernelGGL((nw_bn_train_fwd_kernel<true, false, true>), dim3((unsigned)c), dim3(thr), 0, st, x, residual, gamma, beta,
                               running_mean, running_var, y, save_mean, save_invstd, num_batches_tracked, n, c, hw, x_batch_stride, momentum, eps);
        NW_CHECK_LAUNCH();
        return NW_OK;
    }
#define NW_BNF(R_, V_)                                                                                           \
    hipLaunchKernelGGL((nw_bn_train_fwd_kernel<R_, V_, false>), dim3((unsigned)c), dim3(thr), 0, st, x, residual, gamma, beta, \
                       running_mean, running_var, y, save_mean, save_invstd, num_batches_tracked, n, c, hw,             \
                       x_batch_stride, momentum, eps)
    if (relu) { if (vec) NW_BNF(true, true); else NW_BNF(true, false); }
    else { if (vec) NW_BNF(false, true); else NW_BNF(false, false); }
#undef NW_BNF
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_bn_relu_train_bwd_f32(const float* x, const float* residual, const float* dy, const float* gamma,
                                        const float* beta, const float* save_mean, const float* save_invstd, float* dx,
                                        float* dresidual, float* dgamma, float* dbeta, const float* acc,
                                        int64_t acc_batch_stride, int64_t n, int64_t c, int64_t hw,
                                        int64_t x_batch_stride, int relu, void* stream) {
    using namespace nw;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n <= 0 || c < 0 || hw <= 0 || x_batch_stride < c * hw || c > 0x7fffffffLL || n * hw > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    if (c == 0) return NW_OK;
    if (!x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta) return NW_ERR_INVALID_ARG;
    const bool vec = hw % 4 == 0 && x_batch_stride % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                       reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(dresidual) |
                       reinterpret_cast<uintptr_t>(acc)) & 15) == 0 && (!acc || acc_batch_stride % 4 == 0);
    if (acc && acc_batch_stride < c * hw) return NW_ERR_INVALID_ARG;
    const unsigned thr = channel_threads(n * hw);
    if ((residual != nullptr) != (dresidual != nullptr) || (residual && !relu)) return NW_ERR_INVALID_ARG;
    if (residual) {
        if (vec)
            hipLaunchKernelGGL((nw_bn_train_bwd_kernel<true, true, true>), dim3((unsigned)c), dim3(thr), 0, st, x, residual, dy, gamma,
                               beta, save_mean, save_invstd, dx, dresidual, dgamma, dbeta, acc, acc_batch_stride, n, c, hw, x_batch_stride);
        else
            hipLaunchKernelGGL((nw_bn_train_bwd_kernel<true, false, true>), dim3((unsigned)c), dim3(thr), 0, st, x, residual, dy, gamma,
                               beta, save_mean, save_invstd, dx, dresidual, dgamma, dbeta, acc, acc_batch_stride, n, c, hw, x_batch_stride);
        NW_CHECK_LAUNCH();
        return NW_OK;
    }
#define NW_BNB(R_, V_)                                                                                          \
    hipLaunchKernelGGL((nw_bn_train_bwd_kernel<R_, V_, false>), dim3((unsigned)c), dim3(thr), 0, st, x, residual, dy, gamma, beta, \
                       save_mean, save_invstd, dx, dresidual, dgamma, dbeta, acc, acc_batch_stride, n, c, hw, x_batch_stride)
    if (relu) { if (vec) NW_BNB(true, true); else NW_BNB(true, false); }
    else { if (vec) NW_BNB(false, true); else NW_BNB(false, false); }
#undef NW_BNB
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_bias_act_nhwc_f32(const float* x, const float* bias, const float* residual, int relu, float* out,
                                    int64_t rows, int64_t c, void* stream) {
    using namespace nw;
    if (rows < 0 || c < 0) return NW_ERR_INVALID_ARG;
    if (rows == 0 || c == 0) return NW_OK;
    if (!x || !bias || !out) return NW_ERR_INVALID_ARG;
    if (c % 4 != 0 || c / 4 > 0x7fffffffLL) return NW_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias) |
         reinterpret_cast<uintptr_t>(residual)) & 15)
        return NW_ERR_INVALID_ARG;
    const int64_t total4 = rows * (c / 4);
    int64_t blocks = (total4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NW_BA(R_, S_)                                                                                                 \
    hipLaunchKernelGGL((nw_bias_act_rows_kernel<R_, S_>), dim3((unsigned)blocks), dim3(256), 0, st, x, bias, residual, out, \
                       total4, (int)(c / 4))
    if (relu) { if (residual) NW_BA(true, true); else NW_BA(true, false); }
    else { if (residual) NW_BA(false, true); else NW_BA(false, false); }
#undef NW_BA
    NW_CHECK_LAUNCH();
    return NW_OK;
}

