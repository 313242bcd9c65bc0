// The optimizer step of the training harness (reference train.py:243-247: torch.optim.SGD with momentum, weight decay and
// Nesterov momentum) for MANY parameter tensors in a few launches: a backbone has hundreds of them (DenseNet-121: 364, two
// thirds of them BatchNorm vectors of <= 1024 floats) and torch's multi-tensor kernels spend ~0.3 ms of a 16 ms step on what is
// 160 MB of traffic.  Up to SGD_MAX tensors travel in the kernel arguments (no table upload, no host synchronisation); a
// workgroup finds its tensor by bisection of the prefix table and updates 4096 elements of it:
//   g = grad + wd p;  buf = first step ? g : mu buf + g;  p -= lr (nesterov ? g + mu buf : buf)          (dampening 0)
#include <hip/hip_runtime.h>
#include <cstdint>
#include "nw_internal.h"
#include "../../include/nwhead_hip.h"

namespace {

constexpr int SGD_MAX = 96;          // tensors per launch: 4 + 97 * 4 + 96 * 28 = 3080 bytes of kernel arguments
constexpr int SGD_CHUNK = 4096;      // elements per workgroup: 256 lanes x 4 float4

struct SgdBatch {
    int n;
    int first[SGD_MAX + 1];          // first workgroup of tensor j; first[n] = the grid
    float* p[SGD_MAX];
    const float* g[SGD_MAX];
    float* b[SGD_MAX];
    int len[SGD_MAX];
};

__global__ __launch_bounds__(256) void nw_sgd_kernel(const SgdBatch bt, float lr, float mu, float wd, int nesterov, int init_buf) {
    int lo = 0, hi = bt.n;           // the tensor whose workgroup range holds blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= bt.first[mid]) lo = mid; else hi = mid;
    }
    float* __restrict__ p = bt.p[lo];
    const float* __restrict__ g = bt.g[lo];
    float* __restrict__ b = bt.b[lo];
    const int n = bt.len[lo];
    const int base = ((int)blockIdx.x - bt.first[lo]) * SGD_CHUNK;
    auto one = [&](float pv, float gv, float bv, float& pn, float& bn) {
        const float gg = __builtin_fmaf(wd, pv, gv);
        bn = (b && mu != 0.f) ? (init_buf ? gg : __builtin_fmaf(mu, bv, gg)) : gg;
        const float upd = (b && mu != 0.f) ? (nesterov ? __builtin_fmaf(mu, bn, gg) : bn) : gg;
        pn = __builtin_fmaf(-lr, upd, pv);
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = base + 4 * (threadIdx.x + 256 * k);
        if (e >= n) break;
        if (vec && e + 4 <= n) {
            const float4 pv = *reinterpret_cast<const float4*>(p + e), gv = *reinterpret_cast<const float4*>(g + e);
            const float4 bv = (b && !init_buf) ? *reinterpret_cast<const float4*>(b + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 pn, bn;
            one(pv.x, gv.x, bv.x, pn.x, bn.x); one(pv.y, gv.y, bv.y, pn.y, bn.y);
            one(pv.z, gv.z, bv.z, pn.z, bn.z); one(pv.w, gv.w, bv.w, pn.w, bn.w);
            *reinterpret_cast<float4*>(p + e) = pn;
            if (b) *reinterpret_cast<float4*>(b + e) = bn;
        } else {
            for (int j = e; j < min(e + 4, n); ++j) {
                float pn, bn;
                one(p[j], g[j], (b && !init_buf) ? b[j] : 0.f, pn, bn);
                p[j] = pn;
                if (b) b[j] = bn;
            }
        }
    }
}

}  // namespace

extern "C" int nw_sgd_step_f32(const nw_sgd_param* params, int64_t nparams, float lr, float momentum, float weight_decay,
                               int nesterov, int init_buf, void* stream) {
    if (nparams < 0 || (nparams > 0 && !params)) return NW_ERR_INVALID_ARG;
    if (nesterov && momentum <= 0.f) return NW_ERR_INVALID_ARG;          // (torch: "Nesterov momentum requires a momentum")
    for (int64_t k = 0; k < nparams; ++k) {
        const nw_sgd_param& q = params[k];
        if (q.n < 0 || q.n >= (1LL << 31) - SGD_CHUNK) return NW_ERR_INVALID_ARG;
        if (q.n > 0 && (!q.param || !q.grad || (momentum != 0.f && !q.momentum_buf))) return NW_ERR_INVALID_ARG;
        if ((reinterpret_cast<uintptr_t>(q.param) | reinterpret_cast<uintptr_t>(q.grad) | reinterpret_cast<uintptr_t>(q.momentum_buf)) & 3)
            return NW_ERR_INVALID_ARG;
    }
    if (nparams == 0) return NW_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    SgdBatch bt;
    bt.n = 0; bt.first[0] = 0;
    auto flush = [&]() {
        if (bt.n) hipLaunchKernelGGL(nw_sgd_kernel, dim3((unsigned)bt.first[bt.n]), dim3(256), 0, st, bt, lr, momentum, weight_decay,
                                     nesterov ? 1 : 0, init_buf ? 1 : 0);
        bt.n = 0; bt.first[0] = 0;
    };
    for (int64_t k = 0; k < nparams; ++k) {
        const nw_sgd_param& q = params[k];
        if (q.n == 0) continue;
        bt.p[bt.n] = q.param; bt.g[bt.n] = q.grad; bt.b[bt.n] = momentum != 0.f ? q.momentum_buf : nullptr; bt.len[bt.n] = (int)q.n;
        bt.first[bt.n + 1] = bt.first[bt.n] + (int)((q.n + SGD_CHUNK - 1) / SGD_CHUNK);
        if (++bt.n == SGD_MAX) flush();
    }
    flush();
    NW_CHECK_LAUNCH();
    return NW_OK;
}
