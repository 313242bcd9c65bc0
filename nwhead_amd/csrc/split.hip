// split.hip -- fp32 rows -> "split rows" for the fp16 matrix cores (gfx950 / MI355X only).
//
// For every row: e = power-of-two exponent that brings the row's largest magnitude into
// [2^13, 2^14); every 32-float chunk of the scaled row becomes [32 x h | 32 x l] (fp16), h = fp16(x),
// l = fp16(x - h): same bytes, same row stride as the fp32 original (format: tile_f16.h).  Also written:
// scale[r] = 2^-e (what undoes the scaling of a dot product) and norm2[r] = sum_k x_k^2 in fp32 of the
// ORIGINAL values (the pow(2).sum(-1) half of torch.cdist's matmul form).
// One wave per row; HBM-bound streaming (8*d bytes per row).
#include "nw_internal.h"
#include <type_traits>
#include <cstdlib>

namespace nw {
namespace {

__global__ __launch_bounds__(256) void nw_split_rows_kernel(const float* __restrict__ x,
                                                             float* __restrict__ out,
                                                             float* __restrict__ scale,
                                                             float* __restrict__ norm2, int64_t rows,
                                                             int64_t d, unsigned lmask) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float4* src = reinterpret_cast<const float4*>(x + r * d);
    const int64_t n4 = d / 4;
    // rows up to 64*4*KEEP floats stay in registers: ONE round of loads (the query batch is split
    // inside every forward call, where this kernel is pure latency)
    constexpr int KEEP = 8;
    const bool in_regs = n4 <= 64 * KEEP;
    float4 keep[KEEP];
    float mx = 0.f, n2 = 0.f;
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            const int64_t c = lane + 64 * u;
            keep[u] = (c < n4) ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            const float4 v = keep[u];
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            n2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else {
        for (int64_t c = lane; c < n4; c += 64) {
            const float4 v = src[c];
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            n2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    }
    mx = wave_max(mx);
    n2 = wave_sum(n2);
    int e = 0;
    if (mx > 0.f && mx < INFINITY) {
        frexpf(mx, &e);   // mx = f * 2^e, f in [0.5, 1)
        e = 14 - e;       // mx * 2^e in [2^13, 2^14)
        if (e > 126) e = 126;  // a row of subnormal magnitudes (max < 2^-112): 2^e must stay finite and 2^-e normal
    }
    const float up = ldexpf(1.f, e);
    if (lane == 0) {
        scale[r] = ldexpf(1.f, -e);
        norm2[r] = n2;
    }
    // chunk c (4 floats) of the row -> halves 4*(c % 8) .. +3 of the 32-k chunk c / 8
    _Float16* dst = reinterpret_cast<_Float16*>(out + r * d);
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    auto emit = [&](int64_t c, const float4 v) {
        const float sv[4] = {v.x * up, v.y * up, v.z * up, v.w * up};
        half4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (_Float16)sv[k];
            l[k] = (_Float16)(sv[k] - (float)h[k]);
        }
        if (lmask != 0xffffu) {  // diagnostic (NW_SPLIT_LBITS): fewer significant bits in the low halves
            typedef unsigned short us4 __attribute__((ext_vector_type(4)));
            us4 b = __builtin_bit_cast(us4, l);
            b &= (unsigned short)lmask;
            l = __builtin_bit_cast(half4, b);
        }
        const int64_t chunk = c >> 3, within = (c & 7) * 4;
        *reinterpret_cast<half4*>(dst + chunk * 64 + within) = h;
        *reinterpret_cast<half4*>(dst + chunk * 64 + 32 + within) = l;
    };
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            const int64_t c = lane + 64 * u;
            if (c < n4) emit(c, keep[u]);
        }
    } else {
        for (int64_t c = lane; c < n4; c += 64) emit(c, src[c]);
    }
}

}  // namespace

int launch_split_rows(const float* x, float* out, float* scale, float* norm2, int64_t rows, int64_t d,
                      hipStream_t st) {
    if (rows <= 0) return NW_OK;
    if ((rows + 3) / 4 > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    // Mantissa bits kept in the low halves (default 10 = all: the split is exact to 22 bits).  The tile kernel's pace
    // follows the operands' bit activity: K3 launch 583 us at 10 bits, 572 at 6, 563 at 3, 556 at 0 (max error vs fp64
    // 0.9 / 1.1 / 3.7 / 24 e-6).  Not taken: the large-norm, small-distance cases (golden G8) need the bits.  (Diagnostic
    // knob "split_lbits".)
    const int kb = knob(KNOB_SPLIT_LBITS);
    const int keep = kb == KNOB_UNSET ? 10 : kb;
    const unsigned lmask = keep >= 10 ? 0xffffu : (0xffffu << (10 - (keep < 0 ? 0 : keep))) & 0xffffu;
    hipLaunchKernelGGL(nw_split_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, out, scale,
                       norm2, rows, d, lmask);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

}  // namespace nw

extern "C" int nw_split_rows_f16x2(const float* x, float* out_split, float* row_scale, float* row_norm2,
                                   int64_t rows, int64_t d, void* stream) {
    if (rows < 0 || d < 0) return NW_ERR_INVALID_ARG;
    if (d % 32 != 0) return NW_ERR_UNSUPPORTED;
    if (rows == 0) return NW_OK;
    if (!x || !out_split || !row_scale || !row_norm2) return NW_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out_split)) & 15) return NW_ERR_INVALID_ARG;
    return nw::launch_split_rows(x, out_split, row_scale, row_norm2, rows, d, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------------------------
// All convolution weights of a network -> the split-row operands of conv_nhwc.hip, in ONE launch per optimizer step
// (the weights change every step; per convolution this was a permute, a flip and two split launches).
// A job reads a torch (Cout, Cin, KH, KW) contiguous weight and writes one operand:
//   mode 0  forward      rows = Cout, row co = [t][ci]               (the channels_last bytes of the weight)
//   mode 1  data grad    rows = Cin,  row ci = [T - 1 - t][co]       (flipped taps, transposed channels)
//   mode 2  few channels rows = Cout, row co = [ky][32: kx * 4 + ci] (ROWRUN4: Cin <= 4 padded to 4, a kernel row per chunk)
// jobs (device, int64 x 10 per job): src address, split offset (floats), scale offset, first row (prefix sum), rows, cols,
// Cin, Cout, T, KW | mode << 32
namespace nw {
namespace {

__global__ __launch_bounds__(256) void nw_split_conv_weights_kernel(const int64_t* __restrict__ jobs, int njobs, int64_t total_rows,
                                                                     float* __restrict__ split_base, float* __restrict__ scale_base) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= total_rows) return;
    int lo = 0, hi = njobs - 1;                     // the job whose row range holds r
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[10 * mid + 3] <= r) lo = mid; else hi = mid - 1;
    }
    const int64_t* jb = jobs + 10 * lo;
    const float* src = reinterpret_cast<const float*>(jb[0]);
    const int row = (int)(r - jb[3]), cols = (int)jb[5], Cin = (int)jb[6], Cout = (int)jb[7], T = (int)jb[8];
    const int KW = (int)(jb[9] & 0xffffffff), mode = (int)(jb[9] >> 32);
    auto at = [&](int k) -> float {
        if (mode == 0) { const int t = k / Cin, ci = k - t * Cin; return src[((int64_t)row * Cin + ci) * T + t]; }
        if (mode == 1) { const int t = k / Cout, co = k - t * Cout; return src[((int64_t)co * Cin + row) * T + (T - 1 - t)]; }
        const int ky = k >> 5, j = k & 31, kx = j >> 2, ci = j & 3;
        return (kx < KW && ci < Cin) ? src[((int64_t)row * Cin + ci) * T + ky * KW + kx] : 0.f;
    };
    // Rows of up to 64 x 72 elements (a 3 x 3 x 512 weight's) stay in registers between the maximum and the split: the gathers of
    // the transposed (data-gradient) form are strided by Cin T floats -- every element its own cache line -- and were made twice
    // (ResNet-18: 308 -> 123 us per step).  Three register tiers, so that a short row does not walk 72 guarded slots.
    _Float16* dst = reinterpret_cast<_Float16*>(split_base + jb[1] + (int64_t)row * cols);
    auto finish = [&](float mx) {
        mx = wave_max(mx);
        int e = 0;
        if (mx > 0.f && mx < INFINITY) {
            frexpf(mx, &e);
            e = 14 - e;
            if (e > 126) e = 126;
        }
        if (lane == 0) scale_base[jb[2] + row] = ldexpf(1.f, -e);
        return ldexpf(1.f, e);
    };
    auto put = [&](int k, float x, float up) {      // element k -> half (k % 32) of the 32-k chunk k / 32; l sits 32 halves on
        const float v = x * up;
        const _Float16 h = (_Float16)v;
        dst[(k >> 5) * 64 + (k & 31)] = h;
        dst[(k >> 5) * 64 + 32 + (k & 31)] = (_Float16)(v - (float)h);
    };
    auto in_registers = [&](auto nrc) {
        constexpr int NR = decltype(nrc)::value;
        float keep[NR];
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int k = lane + 64 * j;
            keep[j] = k < cols ? at(k) : 0.f;
            mx = fmaxf(mx, fabsf(keep[j]));
        }
        const float up = finish(mx);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int k = lane + 64 * j;
            if (k < cols) put(k, keep[j], up);
        }
    };
    if (cols <= 64 * 8) in_registers(std::integral_constant<int, 8>{});
    else if (cols <= 64 * 24) in_registers(std::integral_constant<int, 24>{});
    else if (cols <= 64 * 72) in_registers(std::integral_constant<int, 72>{});
    else {
        float mx = 0.f;
        for (int k = lane; k < cols; k += 64) mx = fmaxf(mx, fabsf(at(k)));
        const float up = finish(mx);
        for (int k = lane; k < cols; k += 64) put(k, at(k), up);
    }
}

}  // namespace
}  // namespace nw

extern "C" int nw_split_conv_weights_f16x2(const int64_t* jobs, int64_t njobs, int64_t total_rows, float* split_base,
                                           float* scale_base, void* stream) {
    if (njobs < 0 || total_rows < 0) return NW_ERR_INVALID_ARG;
    if (njobs == 0 || total_rows == 0) return NW_OK;
    if (!jobs || !split_base || !scale_base || njobs > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    hipLaunchKernelGGL(nw::nw_split_conv_weights_kernel, dim3((unsigned)((total_rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), jobs, (int)njobs, total_rows, split_base, scale_base);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
