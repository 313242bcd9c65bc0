// bnrelu.hip -- eval-mode BatchNorm (+ ReLU) of the pre-activation backbones as ONE pass
// (gfx950 / MI355X only).
//
// DenseNet applies BatchNorm -> ReLU -> conv (reference model/densenet.py:33-60, :82-91, :139): the
// BatchNorm cannot be folded into a convolution, but at inference it is y = max(a_c x + b_c, 0) with
// a_c = gamma / sqrt(var + eps), b_c = beta - mean a_c; torch runs it as two kernels (batch norm, then
// relu), each a full read + write of the activation.  The input may be the first c channels of a wider
// slab (the concat-free dense block): rows (n, c) are hw contiguous floats, batch stride given.
#include "nw_internal.h"

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
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(out + plane * hw + 4 * within) = v;
        } else {
            float v = __builtin_fmaf(x[n * x_batch_stride + c * hw + within], a, b);
            if (RELU) v = fmaxf(v, 0.f);
            out[plane * hw + within] = v;
        }
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
