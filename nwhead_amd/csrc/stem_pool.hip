// stem_pool.hip -- the ImageNet stems at inference in ONE kernel (gfx950 / MI355X only; round 4):
//
//     y = maxpool3x3/2/1( relu( conv7x7/2/3(x) + bias ) )          model/resnet.py:147, :200-203; model/densenet.py:114-120
//
// (the BatchNorm behind the convolution folded into weight and bias).  As two kernels -- conv_nhwc.hip's few-channel mode, then
// pool_nhwc.hip -- the 64-channel 112 x 112 map crosses HBM twice (write 205 MB, read 205 MB for 64 images) around 15 GFLOP of
// arithmetic: 172 + 54 us of K2's 1.32 ms.  Here it never leaves the CU: a workgroup owns a 4 x 12 tile of POOLED pixels,
// computes the 9 x 25 convolution outputs under it (18 % of them twice, in neighbouring tiles) into LDS and writes the maxima.
//
// Arithmetic: conv_nhwc.hip's (split-fp16 operands x = h + l, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate).  The
// input is the 4-channel padded image of nw_to_nhwc_pad_f32 (one pixel = one float4) with its amax record; the weight operand is
// the few-channel form of nw_split_rows_f16x2 / nw_split_conv_weights_f16x2: rows (64, 7 x 32), k = 32 ky + 4 kx + ci -- one
// kernel ROW is one 32-wide k chunk, and the eight k values a lane holds (4 kx + ci, kx = 2 g, 2 g + 1) are two neighbouring
// input pixels: 16 bytes of the split patch in LDS, no gather.
//
// Persistent 512-thread workgroups: every wave keeps the weight fragments of ITS 32 channels in registers (112 of them, loaded
// once); per tile the input patch (23 x 56 pixels: requested one tile ahead, in registers) is split into an h and an l plane in
// LDS, eight waves multiply four 16-pixel blocks x 32 channels each (7 k chunks), the results pass through bias + ReLU into a (225, 64) LDS map -- pixels
// outside the image as 0, which under a ReLU is the max pool's own padding -- and 48 x 16 float4 maxima go out, with the amax
// record of what was written.
#include "nw_internal.h"
#include "tile_dma.h"
#include <cstdlib>

namespace nw {
namespace {

constexpr int SP_TPH = 4, SP_TPW = 12;                 // pooled pixels per tile
constexpr int SP_CH = 2 * SP_TPH + 1, SP_CW = 2 * SP_TPW + 1;   // convolution outputs under them: 9 x 25
constexpr int SP_NPX = SP_CH * SP_CW;                  // 225 (16 blocks of 16: 256 slots)
constexpr int SP_PR = 2 * SP_CH + 5, SP_PC = 2 * SP_CW + 6;     // input patch: 23 x 56 pixels of 4 channels
constexpr int SP_PB = SP_PR * SP_PC * 8;               // one plane (h or l) of the patch: 4 halves per pixel
constexpr int SP_OS = 272;                             // bytes between two pixels of the result map (64 floats + 16: fewer bank conflicts)
constexpr int SP_LDS = 2 * SP_PB + SP_NPX * SP_OS;
constexpr int SP_NLD = (SP_PR * SP_PC + 511) / 512;    // patch pixels per thread
static_assert(SP_LDS <= 160 * 1024 - 256, "LDS");

struct StemP {
    const float* x; const float* amax_in;
    const char* ws; const float* wscale; const float* bias;
    float* y; float* amax_out;
    int N, H, W, Ho, Wo, Hp, Wp, ldy, tx, ty;
};

__global__ __launch_bounds__(512, 1) void nw_stem_pool_kernel(const StemP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ph = smem;
    char* const pl = ph + SP_PB;
    char* const om = pl + SP_PB;
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;

    float amx;
    {
        const float4 v = reinterpret_cast<const float4*>(p.amax_in)[lane];
        amx = wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    const int ex = split_exponent(amx);
    const float up = __builtin_ldexpf(1.f, ex), inv_up = __builtin_ldexpf(1.f, -ex);
    // This wave's share: channel blocks 2 hc, 2 hc + 1 (32 channels) x pixel blocks 4 pq .. 4 pq + 3 (64 convolution outputs).  Its
    // weight fragments -- 2 blocks x 7 kernel rows x (h, l): 112 registers -- are loaded ONCE and stay: no weight image in LDS, and
    // the fragment traffic of a tile is the activations' alone (as a 56 KB LDS image read by all eight waves it was 2/3 of it)
    const int hc = wave & 1, pq = wave >> 1;
    half8 ah[2][7], al[2][7];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const char* src = p.ws + (size_t)(16 * (2 * hc + a) + i) * 896 + ky * 128 + g * 16;
            ah[a][ky] = *reinterpret_cast<const half8*>(src);
            al[a][ky] = *reinterpret_cast<const half8*>(src + 64);
        }
    const int per_img = p.tx * p.ty, ntiles = p.N * per_img;
    auto origin = [&](int t, int& n, int& py0, int& px0) {
        n = t / per_img;
        const int r = t - n * per_img, tyi = r / p.tx;
        py0 = tyi * SP_TPH;
        px0 = (r - tyi * p.tx) * SP_TPW;
    };
    // patch pixel k of this thread (k = tid + 512 j) of tile t: the float4 of input pixel (4 py0 - 5 + r, 4 px0 - 5 + c), zeros outside
    float4 pre[SP_NLD];
    auto request = [&](int t) {
        int n, py0, px0;
        origin(t, n, py0, px0);
#pragma unroll
        for (int j = 0; j < SP_NLD; ++j) {
            const int k = tid + 512 * j;
            const int r = k / SP_PC, c = k - r * SP_PC;
            const int iy = 4 * py0 - 5 + r, ix = 4 * px0 - 5 + c;
            const bool ok = k < SP_PR * SP_PC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            pre[j] = ok ? *reinterpret_cast<const float4*>(p.x + (((size_t)n * p.H + iy) * p.W + ix) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // this wave's four pixel blocks: convolution outputs px = 64 pq + 16 b + i of the tile (slots past 224: a dummy pixel 0)
    int boff[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int px = 64 * pq + 16 * b + i;
        const int q = px < SP_NPX ? px : 0, cy = q / SP_CW, cx = q - cy * SP_CW;
        boff[b] = ((2 * cy) * SP_PC + 2 * cx + 2 * g) * 8;          // + ky SP_PC 8 per kernel row
    }
    // the lane's per-channel factors (channels 16 a + 4 g .. + 3), once: requested per tile they cost an L2 round trip each
    float4 sc4[2], bi4[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int co = 16 * (2 * hc + a) + 4 * g;
        const float4 s4 = *reinterpret_cast<const float4*>(p.wscale + co);
        sc4[a] = make_float4(s4.x * inv_up, s4.y * inv_up, s4.z * inv_up, s4.w * inv_up);
        bi4[a] = *reinterpret_cast<const float4*>(p.bias + co);
    }
    float mx = 0.f;
    int t = blockIdx.x;
    if (t < ntiles) request(t);
    for (; t < ntiles; t += gridDim.x) {
        int n, py0, px0;
        origin(t, n, py0, px0);
        __syncthreads();                                          // (the previous tile's pool phase has read the maps; the weights are staged)
#pragma unroll
        for (int j = 0; j < SP_NLD; ++j) {
            const int k = tid + 512 * j;
            if (k < SP_PR * SP_PC) {
                const float v[4] = {pre[j].x * up, pre[j].y * up, pre[j].z * up, pre[j].w * up};
                typedef _Float16 half4_ __attribute__((ext_vector_type(4)));
                half4_ h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = (_Float16)v[e];
                    l[e] = (_Float16)(v[e] - (float)h[e]);
                }
                *reinterpret_cast<half4_*>(ph + k * 8) = h;
                *reinterpret_cast<half4_*>(pl + k * 8) = l;
            }
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) request(t + gridDim.x);  // the next tile's patch rides under this tile's arithmetic
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        half8 bh[1][4], bl[1][4];
        auto frags = [&](int set, int ky) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                bh[set][b] = *reinterpret_cast<const half8*>(ph + boff[b] + ky * SP_PC * 8);
                bl[set][b] = *reinterpret_cast<const half8*>(pl + boff[b] + ky * SP_PC * 8);
                if (g == 3) {          // kx = 7 does not exist: its weights are zeros, but 0 x NaN / inf of the pixel next door is not
#pragma unroll
                    for (int e = 4; e < 8; ++e) { bh[set][b][e] = (_Float16)0.f; bl[set][b][e] = (_Float16)0.f; }
                }
            }
        };
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            constexpr int s_ = 0;
            frags(0, ky);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a][ky], bh[s_][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a][ky], bl[s_][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a][ky], bh[s_][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // bias + ReLU into the result map; a convolution output outside the image counts as 0 (<= every ReLU output: the pool's padding)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int px = 64 * pq + 16 * b + i;
            if (px < SP_NPX) {
                const int cy = px / SP_CW, cx = px - cy * SP_CW;
                const int ay = 2 * py0 - 1 + cy, ax = 2 * px0 - 1 + cx;
                const bool in = ay >= 0 && ay < p.Ho && ax >= 0 && ax < p.Wo;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int co = 16 * (2 * hc + a) + 4 * g;
                    float4 v;
                    v.x = __builtin_fmaf(acc[a][b][0], sc4[a].x, bi4[a].x); v.y = __builtin_fmaf(acc[a][b][1], sc4[a].y, bi4[a].y);
                    v.z = __builtin_fmaf(acc[a][b][2], sc4[a].z, bi4[a].z); v.w = __builtin_fmaf(acc[a][b][3], sc4[a].w, bi4[a].w);
                    v.x = (in && !(v.x < 0.f)) ? v.x : 0.f; v.y = (in && !(v.y < 0.f)) ? v.y : 0.f;      // (!(v < 0): keeps a NaN, like torch's relu)
                    v.z = (in && !(v.z < 0.f)) ? v.z : 0.f; v.w = (in && !(v.w < 0.f)) ? v.w : 0.f;
                    *reinterpret_cast<float4*>(om + px * SP_OS + co * 4) = v;
                }
            }
        }
        __syncthreads();
        // the 3 x 3 / 2 maxima: pooled pixel (r, c) of the tile reads result rows 2 r .. 2 r + 2, columns 2 c .. 2 c + 2
        for (int k = tid; k < SP_TPH * SP_TPW * 16; k += 512) {
            const int q = k >> 4, cq = k & 15;
            const int r = q / SP_TPW, c = q - r * SP_TPW;
            const int py = py0 + r, pxx = px0 + c;
            if (py < p.Hp && pxx < p.Wp) {
                float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
                bool nan = false;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float4 v = *reinterpret_cast<const float4*>(om + ((2 * r + dy) * SP_CW + 2 * c + dx) * SP_OS + cq * 16);
                        nan = nan || v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w;
                        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                    }
                if (nan) {                                         // torch's max pool propagates a NaN; fmaxf drops it: redo, keeping them
                    m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const float4 v = *reinterpret_cast<const float4*>(om + ((2 * r + dy) * SP_CW + 2 * c + dx) * SP_OS + cq * 16);
                            m.x = (v.x != v.x || v.x > m.x) ? v.x : m.x; m.y = (v.y != v.y || v.y > m.y) ? v.y : m.y;
                            m.z = (v.z != v.z || v.z > m.z) ? v.z : m.z; m.w = (v.w != v.w || v.w > m.w) ? v.w : m.w;
                        }
                }
                mx = fmaxf(mx, fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w)));
                *reinterpret_cast<float4*>(p.y + (((size_t)n * p.Hp + py) * p.Wp + pxx) * p.ldy + 4 * cq) = m;
            }
        }
    }
    mx = block_max(mx, red);
    if (p.amax_out && tid < 256 && tid % (int)gridDim.x == (int)blockIdx.x) p.amax_out[tid] = tid == (int)blockIdx.x ? mx : 0.f;
}

inline int sp_num_cus() {
    static const int v = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return v;
}

}  // namespace
}  // namespace nw

extern "C" int nw_stem7x7s2_relu_maxpool_supported(int64_t n, int64_t H, int64_t W, int64_t cout) {
    if (n <= 0 || H < 7 || W < 7 || cout != 64) return 0;
    if (n * H * W * 4 >= (1LL << 31)) return 0;
    return 1;
}

extern "C" int nw_stem7x7s2_relu_maxpool_f16x2(const float* x4, const float* amax_in, const float* w_split, const float* w_scale,
                                               const float* bias, float* y, float* amax_out, int64_t n, int64_t H, int64_t W, int64_t ldy,
                                               void* stream) {
    using namespace nw;
    if (n < 0) return NW_ERR_INVALID_ARG;
    if (n == 0) return NW_OK;
    if (!nw_stem7x7s2_relu_maxpool_supported(n, H, W, 64)) return NW_ERR_UNSUPPORTED;
    if (!x4 || !amax_in || !w_split || !w_scale || !bias || !y) return NW_ERR_INVALID_ARG;
    if (ldy == 0) ldy = 64;
    if (ldy < 64 || ldy % 4) return NW_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x4) | reinterpret_cast<uintptr_t>(amax_in) | reinterpret_cast<uintptr_t>(w_split) |
         reinterpret_cast<uintptr_t>(w_scale) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(y) |
         reinterpret_cast<uintptr_t>(amax_out)) & 15)
        return NW_ERR_INVALID_ARG;
    StemP p;
    p.x = x4; p.amax_in = amax_in; p.ws = reinterpret_cast<const char*>(w_split); p.wscale = w_scale; p.bias = bias; p.y = y;
    p.amax_out = amax_out;
    p.N = (int)n; p.H = (int)H; p.W = (int)W;
    p.Ho = (int)((H + 6 - 7) / 2 + 1); p.Wo = (int)((W + 6 - 7) / 2 + 1);
    p.Hp = (p.Ho + 2 - 3) / 2 + 1; p.Wp = (p.Wo + 2 - 3) / 2 + 1;
    if ((int64_t)n * p.Hp * p.Wp * ldy >= (1LL << 31)) return NW_ERR_UNSUPPORTED;
    p.ldy = (int)ldy;
    p.ty = (p.Hp + SP_TPH - 1) / SP_TPH; p.tx = (p.Wp + SP_TPW - 1) / SP_TPW;
    const int64_t tiles = (int64_t)n * p.ty * p.tx;
    int64_t grid = sp_num_cus() < 256 ? sp_num_cus() : 256;        // (<= the amax record's 256 slots)
    if (grid > tiles) grid = tiles;
    static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(nw_stem_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 SP_LDS) == hipSuccess;
    if (!attr) return NW_ERR_LAUNCH;
    hipLaunchKernelGGL(nw_stem_pool_kernel, dim3((unsigned)grid), dim3(512), SP_LDS, static_cast<hipStream_t>(stream), p);
    NW_CHECK_LAUNCH();
    return NW_OK;
}
