// bn_dgrad.hip -- backward of `norm1 -> relu1 -> conv1` of a dense layer (model/densenet.py:36-40, the 1x1 bottleneck) in
// two streaming passes, without the data gradient of conv1 ever reaching memory (gfx950 / MI355X only; round 4).
//
//     g[p, ch]  = sum_j du[p, j] W[j, ch]                       (data gradient of the 1x1 convolution, K = bn_size * growth <= 128)
//     gd        = g where the forward's own (x - mean) a + beta was positive, else 0          (ReLU)
//     dbeta[ch] = sum_p gd,  dgamma[ch] = sum_p gd xhat,  xhat = (x - mean) invstd           (BatchNorm, batch statistics)
//     G[p, ch] += a (gd - dbeta / M - xhat dgamma / M)                                          (into the gradient slab, in place)
//
// The unfused sequence (conv_nhwc.hip data gradient -> bn_nhwc.hip statistics -> apply) moves 8 c floats per pixel and layer
// (write g; read g, x; read g, x, G, write G; the weight gradient's x) -- O(L^2) bytes over a dense block, 25 % of a
// DenseNet-121 training step.  Here g exists only in the MFMA accumulators: pass 1 reads du and x and leaves the two sums, pass 2
// reads du, x, G and writes G: 4 c + 2 K floats.  The 1x1 product costs 3 x 128 fp16 MFMA-MACs per output against 12-16 bytes
// of HBM traffic: both passes are HBM-bound by a factor of ~5, so the product is simply done twice.
//
// One 512-thread workgroup per CU, persistent: it stages the split-fp16 weight rows of ITS channel group (<= 256 channels x
// K: <= 128 KB of LDS, as [k chunk][row][128 B] with conv_nhwc.hip's 16-byte-slot swizzle) and the per-channel factors once,
// then its eight waves walk 16-pixel strips independently -- no barrier in the loop.  A wave's work is one stream of steps
// (strip, 32 channels): operand B = the strip's du rows, split in registers with the tensor's power-of-two scale from the amax
// record (requested during the previous strip's last step); per step the weight fragments come from LDS, the arithmetic is
// conv_nhwc.hip's (al bh + ah bl + ah bh, fp32 accumulate, the weight row's scale undone per channel), and the epilogue works on
// x / G values requested two steps ahead into a ring of three register sets.  Lane (i, g) of a 16 x 16 result block holds
// channels 4 g .. 4 g + 3 of pixel i: 16-byte accesses to x and G.  Sums: DPP row sums over the strip's 16 pixels, per-wave
// accumulators in LDS (ds_add_f32 on addresses only that wave writes: deterministic), one group per workgroup (<= 256 groups for
// bn_nhwc.hip's finalize); no float atomics on shared addresses.
//
// STATUS (round 4): correct (tests/test_conv_nhwc_gpu.py::test_fused_norm1_backward_against_fp64) but NOT the default: on K4 it is
// 15.3 vs 14.5 ms.  rocprofv3 counters (tools/bd_pmc.sh, 131 712 x 224): no over-fetch (297 + 115 MB against 303 + 118 MB
// algorithmic in pass 2), pass 2 at 3.4 TB/s, pass 1 at 2.2 TB/s and instruction-bound (270 VALU per step, a third of them the DPP
// sums, at two waves per SIMD); the 14 x 14 / 7 x 7 layers (40 of 58) take 20-25 us per pass whatever their size.  DESIGN.md 4.7h'.
#include "nw_internal.h"
#include "tile_dma.h"
#include <cstdlib>
#include <type_traits>

namespace nw {
namespace {

constexpr int BD_CG = 256;          // channels per workgroup (4 chunks of 64): 128 KB of weight rows at K = 128
constexpr int BD_SLOTS = 256;       // = NW_AMAX_SLOTS

struct BdP {
    const float* du; const float* amax_du;
    const char* ws; const float* wscale;
    const float* x; int ldx;
    const float* tab; int tc;       // mean | a | beta, rows tc floats apart
    const float* invstd;
    const float* kk;                // pass 2: mean(gd) | mean(gd xhat), C floats apart
    float* G; int ldg;
    float* amax_out;                // pass 2 (nullable)
    float* part;                    // pass 1: [(k * GX + gx) * C + ch], k = 0 sum gd, 1 sum gd xhat
    int M, C, K, GX, cgs;
};

__device__ __forceinline__ float row16_sum(float x) {   // sum over the 16 lanes of a DPP row
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));  // row_ror:8
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));  // row_ror:4
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    return x;
}

template <bool APPLY>
__global__ __launch_bounds__(512, 1) void nw_bn_dgrad1x1_kernel(const BdP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int gx = blockIdx.x, gy = blockIdx.y;
    const int cb = gy * p.cgs, cn = min(p.cgs, p.C - cb);          // this group's channels [cb, cb + cn), cn % 32 == 0
    const int nq = p.K >> 5, rowb = nq * 128;                      // k chunks; bytes of a weight row
    const int qst = p.cgs * 128;                                   // LDS image: [k chunk][row][128 B] (rows 128 B apart: conflict-free reads)
    char* const wl = smem;
    float* const pm = reinterpret_cast<float*>(smem + (size_t)p.cgs * rowb);     // [7][cgs]: mean, a, beta, invstd (pass 2: a invstd k2), scale, (a k1)
    float* const wsum = pm + 7 * p.cgs;                                          // pass 1: [8 waves][2][cgs]
    __shared__ float red[8];

    // ---- prologue: the tensor's scale, the group's weight rows (swizzled), the per-channel factors
    float amx;
    {
        const float4 v = reinterpret_cast<const float4*>(p.amax_du)[lane];
        amx = wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    const int ex = split_exponent(amx);
    const float up = __builtin_ldexpf(1.f, ex), inv_up = __builtin_ldexpf(1.f, -ex);
    {
        const int upr = nq * 8;                                    // 16-byte units per row
        const char* src = p.ws + (size_t)cb * rowb;
        const int total = cn * upr;
        for (int f0 = 0; f0 < total; f0 += 8 * 512) {              // eight loads in flight per thread, then their stores
            float4 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int f = min(f0 + r * 512 + tid, total - 1);
                v[r] = *reinterpret_cast<const float4*>(src + (size_t)f * 16);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int f = f0 + r * 512 + tid;
                if (f < total) {
                    const int row = f / upr, u = f - row * upr, q = u >> 3, slot = u & 7;
                    *reinterpret_cast<float4*>(wl + q * qst + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) = v[r];
                }
            }
        }
        for (int t = tid; t < cn; t += 512) {
            const int ch = cb + t;
            pm[t] = p.tab[ch];
            pm[p.cgs + t] = p.tab[p.tc + ch];
            pm[2 * p.cgs + t] = p.tab[2 * p.tc + ch];
            const float a_ = p.tab[p.tc + ch], is_ = p.invstd[ch];
            pm[4 * p.cgs + t] = p.wscale[ch] * inv_up;
            if (APPLY) {   // dx = a (gd - k1 - xhat k2) = a gd - a k1 - (x - mean) (a invstd k2)
                pm[3 * p.cgs + t] = a_ * is_ * p.kk[p.C + ch];
                pm[5 * p.cgs + t] = a_ * p.kk[ch];
            } else {
                pm[3 * p.cgs + t] = is_;
            }
        }
        if (!APPLY)
            for (int t = tid; t < 16 * p.cgs; t += 512) wsum[t] = 0.f;
    }
    __syncthreads();

    // ---- the wave's work as ONE stream of steps: step = (strip, 32 channels = two 16 x 16 result blocks).  The x / G values of a
    // step are requested two steps ahead into a ring of three register sets (slot = step % 3: the loop is unrolled by three), across
    // strip boundaries, and the next strip's du rows during the current strip's last step: with two waves per SIMD what decides
    // whether the passes reach the HBM rate is that every wave ALWAYS has ~8 KB in flight, not how much it has at its best.
    const int nstrips = (p.M + 15) >> 4;
    const int nsp = cn >> 5;                                       // steps per strip
    const int s0 = gx * 8 + wave, sstride = p.GX * 8;
    const int ns = s0 < nstrips ? (nstrips - 1 - s0) / sstride + 1 : 0;
    const int T = ns * nsp;
    float* const mysum = wsum + wave * 2 * p.cgs;
    const int asw = (i >> 1) & 7;
    const int aoff_h = i * 128 + ((g ^ asw) << 4), aoff_l = i * 128 + (((4 + g) ^ asw) << 4);
    float mx = 0.f;
    auto pixel = [&](int strip) { return min(16 * strip + i, p.M - 1); };

    float4 X[3][2], Gv[3][2];
    int r_strip = s0, r_step = 0, r_t = 0;                         // request cursor
    size_t r_xo = (size_t)pixel(s0) * p.ldx + cb + 4 * g, r_go = (size_t)pixel(s0) * p.ldg + cb + 4 * g;
    auto request = [&](auto slotc) {
        constexpr int slot = decltype(slotc)::value;
        if (r_t < T) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                X[slot][a] = *reinterpret_cast<const float4*>(p.x + r_xo + 32 * r_step + 16 * a);
                if (APPLY) Gv[slot][a] = *reinterpret_cast<const float4*>(p.G + r_go + 32 * r_step + 16 * a);
            }
            ++r_t;
            if (++r_step == nsp) {
                r_step = 0;
                r_strip += sstride;
                const int px = pixel(r_strip);
                r_xo = (size_t)px * p.ldx + cb + 4 * g;
                r_go = (size_t)px * p.ldg + cb + 4 * g;
            }
        }
    };
    float4 dv[4][2];
    auto request_du = [&](int strip) {
        const float* src = p.du + (size_t)pixel(strip) * p.K + 8 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) {
                dv[q][0] = *reinterpret_cast<const float4*>(src + 32 * q);
                dv[q][1] = *reinterpret_cast<const float4*>(src + 32 * q + 4);
            }
    };
    if (ns > 0) request_du(s0);
    request(std::integral_constant<int, 0>{});
    request(std::integral_constant<int, 1>{});

    half8 bh[4], bl[4];
    int c_strip = s0, c_step = 0, t = 0;                           // compute cursor
    bool ok = false;
    size_t c_go = 0;
    auto step = [&](auto slotc) {
        constexpr int slot = decltype(slotc)::value;
        if (t >= T) return;
        request(std::integral_constant<int, (slot + 2) % 3>{});
        if (c_step == 0) {                                         // a new strip: operand B = its du rows, split h + l at the tensor's scale
            ok = 16 * c_strip + i < p.M;
            c_go = (size_t)pixel(c_strip) * p.ldg + cb + 4 * g;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < nq) {
                    const float v[8] = {dv[q][0].x, dv[q][0].y, dv[q][0].z, dv[q][0].w, dv[q][1].x, dv[q][1].y, dv[q][1].z, dv[q][1].w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float sv = ok ? v[k] * up : 0.f;
                        const _Float16 h = (_Float16)sv;
                        bh[q][k] = h;
                        bl[q][k] = (_Float16)(sv - (float)h);
                    }
                }
        }
        if (c_step == nsp - 1 && c_strip + sstride < nstrips) request_du(c_strip + sstride);
        const int t0 = 32 * c_step;                                // first channel of the step inside the group
        constexpr int NF = APPLY ? 6 : 5;                          // mean, a, beta, invstd | a invstd k2, scale [, a k1]
        f32x4 PM[2][NF];
        auto consts = [&](int set_, int tt) {
#pragma unroll
            for (int k = 0; k < NF; ++k) PM[set_][k] = *reinterpret_cast<const f32x4*>(pm + k * p.cgs + tt);
        };
        consts(0, t0 + 4 * g);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const char* wb = wl + t0 * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) {
                half8 ah[2], al[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    ah[a] = *reinterpret_cast<const half8*>(wb + q * qst + (16 * a) * 128 + aoff_h);
                    al[a] = *reinterpret_cast<const half8*>(wb + q * qst + (16 * a) * 128 + aoff_l);
                }
#ifndef BD_ABL_NOMFMA
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[q], acc[a], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[q], acc[a], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[q], acc[a], 0, 0, 0);
#endif
            }
#ifndef BD_ABL_NOEPI
        // ---- epilogue: acc[a][e] = g of the strip's pixel i, channel cb + t0 + 16 a + 4 g + e (before the row scale)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int tt = t0 + 16 * a + 4 * g;
            const f32x4 (&F)[NF] = PM[a];
            if (a == 0) consts(1, tt + 16);
            const float xv[4] = {X[slot][a].x, X[slot][a].y, X[slot][a].z, X[slot][a].w};
            if constexpr (APPLY) {
                const float go[4] = {Gv[slot][a].x, Gv[slot][a].y, Gv[slot][a].z, Gv[slot][a].w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xm = xv[e] - F[0][e];
                    const float pre = __builtin_fmaf(xm, F[1][e], F[2][e]);   // the forward's own value: same ReLU mask
                    const float gd = pre > 0.f ? acc[a][e] * F[4][e] : 0.f;
                    o[e] = __builtin_fmaf(F[1][e], gd, go[e] - F[NF - 1][e]) - xm * F[3][e];
                }
                if (ok) {
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                    *reinterpret_cast<float4*>(p.G + c_go + t0 + 16 * a) = make_float4(o[0], o[1], o[2], o[3]);
                }
            } else {
                float s1[4], s2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xm = xv[e] - F[0][e];
                    const float pre = __builtin_fmaf(xm, F[1][e], F[2][e]);
                    const float gd = (ok && pre > 0.f) ? acc[a][e] * F[4][e] : 0.f;
                    s1[e] = row16_sum(gd);
                    s2[e] = row16_sum(gd * (xm * F[3][e]));
                }
                if (i == 0) {                                      // this wave's running sums: its own LDS rows, one writer per address,
#pragma unroll                                                     //  program order -- ds_add_f32 without a return value: nothing to wait for
                    for (int e = 0; e < 4; ++e) {
                        __hip_atomic_fetch_add(mysum + tt + e, s1[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(mysum + p.cgs + tt + e, s2[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
#else
        if (acc[0][0] == 12345.f) mx = X[slot][0].x + (APPLY ? Gv[slot][0].x : 0.f);
#endif
        ++t;
        if (++c_step == nsp) { c_step = 0; c_strip += sstride; }
    };
    while (t < T) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
    }
    if (APPLY) {
        mx = block_max(mx, red);
        // slots b, b + n, ... of the record belong to workgroup b of n: its maximum in the first, zeros in the others
        if (p.amax_out) {
            const int n = gridDim.x * gridDim.y, b = gy * gridDim.x + gx;
            if (tid < BD_SLOTS && tid % n == b) p.amax_out[tid] = tid == b ? mx : 0.f;
        }
    } else {
        __syncthreads();
        for (int t = tid; t < cn; t += 512) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {                          // fixed order: deterministic
                a1 += wsum[w * 2 * p.cgs + t];
                a2 += wsum[w * 2 * p.cgs + p.cgs + t];
            }
            p.part[((size_t)0 * p.GX + gx) * p.C + cb + t] = a1;
            p.part[((size_t)1 * p.GX + gx) * p.C + cb + t] = a2;
        }
    }
}

inline int num_cus_bd() {
    static const int v = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return v;
}
struct BdPlan { int GX, GY, cgs; size_t lds; };
inline bool bd_plan(int64_t rows, int64_t c, int64_t k, BdPlan* pl) {
    if (rows <= 0 || c < 32 || c % 32 || k < 32 || k > 128 || k % 32) return false;
    const int GY = (int)((c + BD_CG - 1) / BD_CG);
    int cgs = (int)(((c + GY - 1) / GY + 63) / 64 * 64);          // equal groups, whole 64-channel chunks
    if ((int64_t)cgs * (GY - 1) >= c) return false;               // (cannot happen for c % 32 == 0; guards the grid)
    const int64_t nstrips = (rows + 15) / 16;
    int64_t GX = num_cus_bd() / GY;
    if (GX < 1) GX = 1;
    if (GX > (nstrips + 7) / 8) GX = (nstrips + 7) / 8;
    if (GX * GY > BD_SLOTS) GX = BD_SLOTS / GY;
    pl->GX = (int)GX; pl->GY = GY; pl->cgs = cgs;
    pl->lds = (size_t)cgs * k * 4 + (size_t)7 * cgs * 4 + (size_t)16 * cgs * 4;   // weights, factor table (6 rows used), per-wave sums
    return pl->lds <= 160 * 1024 - 64;
}

}  // namespace
}  // namespace nw

extern "C" size_t nw_bn_dgrad1x1_workspace_bytes(int64_t rows, int64_t c) {
    if (rows <= 0 || c <= 0) return 0;
    return ((size_t)2 * nw::BD_SLOTS * c + 2 * c) * sizeof(float);
}

extern "C" int nw_bn_dgrad1x1_bwd_f16x2(const float* du, const float* amax_du, const float* w_split, const float* w_scale,
                                        const float* x, int64_t ldx, const float* tab, int64_t tab_stride, const float* invstd,
                                        float* g, int64_t ldg, float* amax_out, float* dgamma, float* dbeta, void* workspace,
                                        size_t workspace_bytes, int64_t rows, int64_t c, int64_t k, void* stream) {
    using namespace nw;
    if (rows < 0 || c <= 0 || k <= 0) return NW_ERR_INVALID_ARG;
    if (rows == 0) return NW_OK;
    BdPlan pl;
    if (!bd_plan(rows, c, k, &pl)) return NW_ERR_UNSUPPORTED;
    if (!du || !amax_du || !w_split || !w_scale || !x || !tab || !invstd || !g || !dgamma || !dbeta) return NW_ERR_INVALID_ARG;
    if (ldx < c || ldg < c || ldx % 4 || ldg % 4 || tab_stride < c || tab_stride % 4) return NW_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(amax_du) | reinterpret_cast<uintptr_t>(w_split) |
         reinterpret_cast<uintptr_t>(w_scale) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(tab) |
         reinterpret_cast<uintptr_t>(invstd) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(amax_out) |
         reinterpret_cast<uintptr_t>(dgamma) | reinterpret_cast<uintptr_t>(dbeta) | reinterpret_cast<uintptr_t>(workspace)) & 15)
        return NW_ERR_INVALID_ARG;
    if (rows * ldx >= (1LL << 31) || rows * ldg >= (1LL << 31) || rows * k >= (1LL << 31)) return NW_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < nw_bn_dgrad1x1_workspace_bytes(rows, c)) return NW_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const bool attr = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(nw_bn_dgrad1x1_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024 - 64) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(nw_bn_dgrad1x1_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024 - 64) == hipSuccess;
    }();
    if (!attr) return NW_ERR_LAUNCH;
    float* part = static_cast<float*>(workspace);
    float* kk = part + (size_t)2 * BD_SLOTS * c;
    BdP p;
    p.du = du; p.amax_du = amax_du; p.ws = reinterpret_cast<const char*>(w_split); p.wscale = w_scale;
    p.x = x; p.ldx = (int)ldx; p.tab = tab; p.tc = (int)tab_stride; p.invstd = invstd; p.kk = kk;
    p.G = g; p.ldg = (int)ldg; p.amax_out = amax_out; p.part = part;
    p.M = (int)rows; p.C = (int)c; p.K = (int)k; p.GX = pl.GX; p.cgs = pl.cgs;
    hipLaunchKernelGGL(nw_bn_dgrad1x1_kernel<false>, dim3((unsigned)pl.GX, (unsigned)pl.GY), dim3(512), pl.lds, st, p);
    NW_CHECK_LAUNCH();
    const int rc = bn_bwd_finalize_groups(part, pl.GX, (int)c, 1.f / (float)rows, dgamma, dbeta, kk, st);
    if (rc != NW_OK) return rc;
    hipLaunchKernelGGL(nw_bn_dgrad1x1_kernel<true>, dim3((unsigned)pl.GX, (unsigned)pl.GY), dim3(512), pl.lds, st, p);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
