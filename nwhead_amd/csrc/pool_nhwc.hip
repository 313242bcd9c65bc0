// The two pools of the backbones' training path over channels-last fp32 activations (include/nwhead_hip.h):
//   2 x 2 / 2 average pool (a DenseNet transition, reference model/densenet.py:83-91: nn.AvgPool2d(2, 2))
//   3 x 3 / 2 pad 1 max pool (the stems, reference model/densenet.py:114 and model/resnet.py:147: nn.MaxPool2d(3, 2, 1))
// forward and backward.  Pure streaming work: one float4 of channels per lane, consecutive lanes on consecutive channels of
// one pixel (coalesced rows), every byte read once (the overlapping windows of the max pool hit in L2).  The max pool keeps
// the winning tap of each window as one byte, so its backward reads gy and a quarter of gy's bytes instead of int64 indices,
// and is a gather (each input pixel looks at the <= 4 windows that cover it): no atomics, deterministic.
// Rows may be strided (ld* >= c floats): a pool can read from / write into a channel prefix of a wider NHWC tensor.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "nw_internal.h"

namespace {

__global__ __launch_bounds__(256) void nw_avgpool2_nhwc_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                              int H, int W, int Ho, int Wo, int q4, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / q4;                    // output pixel (n, oy, ox)
        const int c = (int)(idx - p * q4) * 4;
        const int ox = (int)(p % Wo);
        const int64_t t = p / Wo;
        const int oy = (int)(t % Ho);
        const int64_t n = t / Ho;
        const float* r0 = x + ((n * H + 2 * oy) * W + 2 * ox) * ldx + c;
        const float* r1 = r0 + (int64_t)W * ldx;
        const float4 a = *reinterpret_cast<const float4*>(r0), b = *reinterpret_cast<const float4*>(r0 + ldx);
        const float4 d = *reinterpret_cast<const float4*>(r1), e = *reinterpret_cast<const float4*>(r1 + ldx);
        // torch's order: the window's sum row by row, then one division by the window size
        *reinterpret_cast<float4*>(y + p * ldy + c) = make_float4((((a.x + b.x) + d.x) + e.x) * 0.25f, (((a.y + b.y) + d.y) + e.y) * 0.25f,
                                                                  (((a.z + b.z) + d.z) + e.z) * 0.25f, (((a.w + b.w) + d.w) + e.w) * 0.25f);
    }
}

__global__ __launch_bounds__(256) void nw_avgpool2_nhwc_bwd_kernel(const float* __restrict__ gy, int64_t ldgy, float* __restrict__ gx,
                                                                  int64_t ldgx, int H, int W, int Ho, int Wo, int q4, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / q4;                    // input pixel (n, iy, ix)
        const int c = (int)(idx - p * q4) * 4;
        const int ix = (int)(p % W);
        const int64_t t = p / W;
        const int iy = (int)(t % H);
        const int64_t n = t / H;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((iy >> 1) < Ho && (ix >> 1) < Wo) {        // an odd last row / column is in no window
            g = *reinterpret_cast<const float4*>(gy + ((n * Ho + (iy >> 1)) * Wo + (ix >> 1)) * ldgy + c);
            g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
        }
        *reinterpret_cast<float4*>(gx + p * ldgx + c) = g;
    }
}

__global__ __launch_bounds__(256) void nw_maxpool3s2_nhwc_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                                unsigned char* __restrict__ tap, int H, int W, int Ho, int Wo, int C,
                                                                int q4, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / q4;
        const int c = (int)(idx - p * q4) * 4;
        const int ox = (int)(p % Wo);
        const int64_t t = p / Wo;
        const int oy = (int)(t % Ho);
        const int64_t n = t / Ho;
        float best[4]; int arg[4];
        bool first = true;
        // torch's scan (MaxPool2d on the GPU): window rows then columns, a later value wins when it is greater or NaN
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                const float4 v4 = *reinterpret_cast<const float4*>(x + ((n * H + iy) * W + ix) * ldx + c);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (first || v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; arg[j] = ky * 3 + kx; }
                first = false;
            }
        }
        *reinterpret_cast<float4*>(y + p * ldy + c) = make_float4(best[0], best[1], best[2], best[3]);
        if (tap)
            *reinterpret_cast<uchar4*>(tap + p * C + c) = make_uchar4((unsigned char)arg[0], (unsigned char)arg[1], (unsigned char)arg[2],
                                                                      (unsigned char)arg[3]);
    }
}

__global__ __launch_bounds__(256) void nw_maxpool3s2_nhwc_bwd_kernel(const float* __restrict__ gy, int64_t ldgy,
                                                                    const unsigned char* __restrict__ tap, float* __restrict__ gx,
                                                                    int64_t ldgx, int H, int W, int Ho, int Wo, int C, int q4, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / q4;                    // input pixel
        const int c = (int)(idx - p * q4) * 4;
        const int ix = (int)(p % W);
        const int64_t t = p / W;
        const int iy = (int)(t % H);
        const int64_t n = t / H;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        // windows with 2 o - 1 <= i <= 2 o + 1: o in [ceil((i - 1) / 2), floor((i + 1) / 2)], in ascending order
        const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;
        for (int oy = oy0; oy <= oy1; ++oy) {
            if (oy >= Ho) break;
            const int ky = iy - (2 * oy - 1);
            for (int ox = ox0; ox <= ox1; ++ox) {
                if (ox >= Wo) break;
                const int mine = ky * 3 + (ix - (2 * ox - 1));
                const int64_t o = (n * Ho + oy) * Wo + ox;
                const uchar4 a = *reinterpret_cast<const uchar4*>(tap + o * C + c);
                if (a.x != mine && a.y != mine && a.z != mine && a.w != mine) continue;
                const float4 v = *reinterpret_cast<const float4*>(gy + o * ldgy + c);
                if (a.x == mine) g[0] += v.x;
                if (a.y == mine) g[1] += v.y;
                if (a.z == mine) g[2] += v.z;
                if (a.w == mine) g[3] += v.w;
            }
        }
        *reinterpret_cast<float4*>(gx + p * ldgx + c) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

inline unsigned pool_grid(int64_t total) {
    const int64_t want = (total + 255) / 256;
    return (unsigned)(want < 1 ? 1 : (want > 65536 ? 65536 : want));
}
inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }
inline bool bad_dims(int64_t n, int64_t h, int64_t w, int64_t c) {
    return n < 0 || h <= 0 || w <= 0 || c <= 0 || c % 4 || h >= (1 << 24) || w >= (1 << 24) || c >= (1 << 24) ||
           n >= (1LL << 31) || n * h * w >= (1LL << 40);
}


// y = avgpool2x2(relu((x - mean) a + beta)): an eval-mode DenseNet transition's BatchNorm + ReLU with the pool in FRONT of its
// bias-free 1 x 1 convolution (the two commute: both linear, per pixel / per channel; model/densenet.py:83-91 runs conv -> pool):
// the convolution then works on a quarter of the pixels.  tab: mean | a | beta (C floats apart).  amax record of y.
__global__ __launch_bounds__(1024) void nw_bn_relu_avgpool2_nhwc_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ tab,
                                                                       float* __restrict__ y, int64_t ldy, float* __restrict__ amax, int H,
                                                                       int W, int Ho, int Wo, int C, int q4, int64_t total) {
    __shared__ float red[16];
    float mx = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / q4;                    // output pixel (n, oy, ox)
        const int c = (int)(idx - p * q4) * 4;
        const int ox = (int)(p % Wo);
        const int64_t t = p / Wo;
        const int oy = (int)(t % Ho);
        const int64_t n = t / Ho;
        const float* r0 = x + ((n * H + 2 * oy) * W + 2 * ox) * ldx + c;
        const float* r1 = r0 + (int64_t)W * ldx;
        const float4 v[4] = {*reinterpret_cast<const float4*>(r0), *reinterpret_cast<const float4*>(r0 + ldx),
                             *reinterpret_cast<const float4*>(r1), *reinterpret_cast<const float4*>(r1 + ldx)};
        const float4 m = *reinterpret_cast<const float4*>(tab + c), a = *reinterpret_cast<const float4*>(tab + C + c),
                     b = *reinterpret_cast<const float4*>(tab + 2 * C + c);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                  // (x < 0 ? 0 : x keeps a NaN, like torch's relu; torch's pool: the sum, then x 0.25)
            const float e0 = __builtin_fmaf(v[k].x - m.x, a.x, b.x), e1 = __builtin_fmaf(v[k].y - m.y, a.y, b.y);
            const float e2 = __builtin_fmaf(v[k].z - m.z, a.z, b.z), e3 = __builtin_fmaf(v[k].w - m.w, a.w, b.w);
            s.x += e0 < 0.f ? 0.f : e0; s.y += e1 < 0.f ? 0.f : e1; s.z += e2 < 0.f ? 0.f : e2; s.w += e3 < 0.f ? 0.f : e3;
        }
        s.x *= 0.25f; s.y *= 0.25f; s.z *= 0.25f; s.w *= 0.25f;
        mx = fmaxf(mx, fmaxf(fmaxf(s.x, s.y), fmaxf(s.z, s.w)));
        *reinterpret_cast<float4*>(y + p * ldy + c) = s;
    }
    mx = nw::block_max(mx, red);
    if (amax && threadIdx.x < 256 && (int)threadIdx.x % (int)gridDim.x == (int)blockIdx.x) amax[threadIdx.x] = threadIdx.x == blockIdx.x ? mx : 0.f;
}

// out = relu(a + b) at the end of a residual block (model/resnet.py:60-66, :100-108), x < 0 ? 0 : x (keeps a NaN, like torch's
// relu), with the amax record of out: slots b, b + n, ... of the record belong to workgroup b of n
__global__ __launch_bounds__(1024) void nw_add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                         float* __restrict__ amax, int64_t n4) {
    __shared__ float red[16];
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        float4 o = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
        o.x = o.x < 0.f ? 0.f : o.x; o.y = o.y < 0.f ? 0.f : o.y; o.z = o.z < 0.f ? 0.f : o.z; o.w = o.w < 0.f ? 0.f : o.w;
        mx = fmaxf(mx, fmaxf(fmaxf(o.x, o.y), fmaxf(o.z, o.w)));
        reinterpret_cast<float4*>(out)[i] = o;
    }
    mx = nw::block_max(mx, red);
    if (amax && threadIdx.x < 256 && (int)threadIdx.x % (int)gridDim.x == (int)blockIdx.x) amax[threadIdx.x] = threadIdx.x == blockIdx.x ? mx : 0.f;
}

// dx = g where out > 0 (the backward of the ReLU above; both summands of the block receive dx), with dx's amax record
__global__ __launch_bounds__(1024) void nw_relu_bwd_kernel(const float* __restrict__ out, const float* __restrict__ g, float* __restrict__ dx,
                                                         float* __restrict__ amax, int64_t n4) {
    __shared__ float red[16];
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 y = reinterpret_cast<const float4*>(out)[i], v = reinterpret_cast<const float4*>(g)[i];
        const float4 o = make_float4(y.x > 0.f ? v.x : 0.f, y.y > 0.f ? v.y : 0.f, y.z > 0.f ? v.z : 0.f, y.w > 0.f ? v.w : 0.f);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        reinterpret_cast<float4*>(dx)[i] = o;
    }
    mx = nw::block_max(mx, red);
    if (amax && threadIdx.x < 256 && (int)threadIdx.x % (int)gridDim.x == (int)blockIdx.x) amax[threadIdx.x] = threadIdx.x == blockIdx.x ? mx : 0.f;
}

}  // namespace

extern "C" int nw_avgpool2x2_nhwc_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n, int64_t h, int64_t w, int64_t c,
                                      void* stream) {
    if (bad_dims(n, h, w, c) || h < 2 || w < 2) return NW_ERR_INVALID_ARG;
    if (ldx == 0) ldx = c;
    if (ldy == 0) ldy = c;
    if (ldx < c || ldx % 4 || ldy < c || ldy % 4) return NW_ERR_INVALID_ARG;
    if (n == 0) return NW_OK;
    if (!x || !y || misaligned(x) || misaligned(y)) return NW_ERR_INVALID_ARG;
    const int Ho = (int)(h / 2), Wo = (int)(w / 2), q4 = (int)(c / 4);
    const int64_t total = n * Ho * Wo * q4;
    hipLaunchKernelGGL(nw_avgpool2_nhwc_kernel, dim3(pool_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, y, ldy,
                       (int)h, (int)w, Ho, Wo, q4, total);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_avgpool2x2_nhwc_bwd_f32(const float* gy, int64_t ldgy, float* gx, int64_t ldgx, int64_t n, int64_t h, int64_t w,
                                          int64_t c, void* stream) {
    if (bad_dims(n, h, w, c) || h < 2 || w < 2) return NW_ERR_INVALID_ARG;
    if (ldgy == 0) ldgy = c;
    if (ldgx == 0) ldgx = c;
    if (ldgy < c || ldgy % 4 || ldgx < c || ldgx % 4) return NW_ERR_INVALID_ARG;
    if (n == 0) return NW_OK;
    if (!gy || !gx || misaligned(gy) || misaligned(gx)) return NW_ERR_INVALID_ARG;
    const int q4 = (int)(c / 4);
    const int64_t total = n * h * w * q4;
    hipLaunchKernelGGL(nw_avgpool2_nhwc_bwd_kernel, dim3(pool_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), gy, ldgy, gx,
                       ldgx, (int)h, (int)w, (int)(h / 2), (int)(w / 2), q4, total);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_maxpool3x3s2_nhwc_f32(const float* x, int64_t ldx, float* y, int64_t ldy, unsigned char* tap, int64_t n, int64_t h,
                                        int64_t w, int64_t c, void* stream) {
    if (bad_dims(n, h, w, c)) return NW_ERR_INVALID_ARG;
    if (ldx == 0) ldx = c;
    if (ldy == 0) ldy = c;
    if (ldx < c || ldx % 4 || ldy < c || ldy % 4) return NW_ERR_INVALID_ARG;
    if (n == 0) return NW_OK;
    if (!x || !y || misaligned(x) || misaligned(y) || (reinterpret_cast<uintptr_t>(tap) & 3)) return NW_ERR_INVALID_ARG;
    const int Ho = (int)((h - 1) / 2 + 1), Wo = (int)((w - 1) / 2 + 1), q4 = (int)(c / 4);   // floor((h + 2 - 3) / 2) + 1
    const int64_t total = n * Ho * Wo * q4;
    hipLaunchKernelGGL(nw_maxpool3s2_nhwc_kernel, dim3(pool_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, y, ldy,
                       tap, (int)h, (int)w, Ho, Wo, (int)c, q4, total);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_maxpool3x3s2_nhwc_bwd_f32(const float* gy, int64_t ldgy, const unsigned char* tap, float* gx, int64_t ldgx, int64_t n,
                                            int64_t h, int64_t w, int64_t c, void* stream) {
    if (bad_dims(n, h, w, c)) return NW_ERR_INVALID_ARG;
    if (ldgy == 0) ldgy = c;
    if (ldgx == 0) ldgx = c;
    if (ldgy < c || ldgy % 4 || ldgx < c || ldgx % 4) return NW_ERR_INVALID_ARG;
    if (n == 0) return NW_OK;
    if (!gy || !gx || !tap || misaligned(gy) || misaligned(gx) || (reinterpret_cast<uintptr_t>(tap) & 3)) return NW_ERR_INVALID_ARG;
    const int Ho = (int)((h - 1) / 2 + 1), Wo = (int)((w - 1) / 2 + 1), q4 = (int)(c / 4);
    const int64_t total = n * h * w * q4;
    hipLaunchKernelGGL(nw_maxpool3s2_nhwc_bwd_kernel, dim3(pool_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), gy, ldgy,
                       tap, gx, ldgx, (int)h, (int)w, Ho, Wo, (int)c, q4, total);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

static int ew_args(const void* a, const void* b, const void* c, const void* amax, int64_t count) {
    if (count < 0 || count % 4) return NW_ERR_INVALID_ARG;
    if (count > 0 && (!a || !b || !c)) return NW_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(amax)) & 15)
        return NW_ERR_INVALID_ARG;
    return NW_OK;
}
static unsigned ew_grid(int64_t n4) {
    const int64_t g = (n4 + 1023) / 1024;
    return (unsigned)(g < 1 ? 1 : (g > 256 ? 256 : g));           // (<= the record's 256 slots: one per workgroup of 1024 threads)
}

extern "C" int nw_add_relu_f32(const float* a, const float* b, float* out, float* amax_out, int64_t count, void* stream) {
    const int rc = ew_args(a, b, out, amax_out, count);
    if (rc != NW_OK) return rc;
    hipLaunchKernelGGL(nw_add_relu_kernel, dim3(ew_grid(count / 4)), dim3(1024), 0, static_cast<hipStream_t>(stream), a, b, out, amax_out, count / 4);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_relu_bwd_f32(const float* out, const float* g, float* dx, float* amax_out, int64_t count, void* stream) {
    const int rc = ew_args(out, g, dx, amax_out, count);
    if (rc != NW_OK) return rc;
    hipLaunchKernelGGL(nw_relu_bwd_kernel, dim3(ew_grid(count / 4)), dim3(1024), 0, static_cast<hipStream_t>(stream), out, g, dx, amax_out, count / 4);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_bn_relu_avgpool2x2_nhwc_f32(const float* x, int64_t ldx, const float* tab, float* y, int64_t ldy, float* amax_out,
                                              int64_t n, int64_t h, int64_t w, int64_t c, void* stream) {
    if (bad_dims(n, h, w, c) || h < 2 || w < 2) return NW_ERR_INVALID_ARG;
    if (ldx == 0) ldx = c;
    if (ldy == 0) ldy = c;
    if (ldx < c || ldx % 4 || ldy < c || ldy % 4) return NW_ERR_INVALID_ARG;
    if (n == 0) return NW_OK;
    if (!x || !y || !tab || misaligned(x) || misaligned(y) || misaligned(tab) || misaligned(amax_out)) return NW_ERR_INVALID_ARG;
    const int Ho = (int)(h / 2), Wo = (int)(w / 2), q4 = (int)(c / 4);
    const int64_t total = n * Ho * Wo * q4;
    hipLaunchKernelGGL(nw_bn_relu_avgpool2_nhwc_kernel, dim3(ew_grid(total)), dim3(1024), 0, static_cast<hipStream_t>(stream), x, ldx, tab, y,
                       ldy, amax_out, (int)h, (int)w, Ho, Wo, (int)c, q4, total);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
