// conv3x3.hip -- 3x3 convolution (stride 1, padding 1) of the backbones' inference path as an implicit GEMM on the
// fp32 matrix cores, bias / ReLU / residual behind it fused in, output written in place into a channel window of a
// wider tensor (a DenseNet block's slab) (gfx950 / MI355X only).
//
// Replaces, in the folded inference copies, the 3x3 convolutions of DenseNet's dense layers (model/densenet.py:41-45:
// conv2, 128 -> 32 channels, 58 per forward, followed by a copy of its output into the block's slab), of
// CIFAR_DenseNet (model/densenet3.py:10-22) and of ResNet's BasicBlock (model/resnet.py:31-66, conv -> folded
// BatchNorm bias -> ReLU [+ identity]).
//
//     out[n, co, y, x] = post( bias[co] + sum_{ci, ky, kx} W[co, ci, ky, kx] * in[n, ci, y + ky - 1, x + kx - 1] [+ res] )
//
// Implicit GEMM: M = cout, N = pixels of ONE image plane (flattened y W + x), K = (ci, tap).  A workgroup computes
// TM output channels x TN consecutive pixels.  Per stage of 8 input channels it needs, for every channel, the plane's
// pixels [p0 - W - 1, p0 + TN + W + 1): ONE contiguous span serves all nine taps -- tap (ky, kx) of pixel p is the span
// element p + (ky - 1) W + (kx - 1) -- so the span goes HBM/L2 -> LDS once (LDS-DMA, 16 bytes per lane when W % 4 == 0)
// and the nine shifted B fragments are plain ds_read_b32 at nine constant offsets.  Image borders cost nothing in the
// loop (fp32 MFMAs do not overlap with vector ALU work on their SIMD: a v_cndmask per loaded value was a third of the
// loop's time): the span lies in LDS with the image's rows `pitch` = W + gap floats apart (gap = 4 with 16-byte DMA
// pieces, else 1) and every piece that is not a pixel of the plane -- the gap behind each row, the rows above the
// first and below the last -- comes from a buffer of ZEROS (the DMA source is chosen per lane).  Tap (ky, kx) of
// pixel (y, x) is LDS element y pitch + x + (ky-1) pitch + (kx-1): column -1 and column W are the gap, rows -1 and H
// are zero rows.  No masks, one accumulator set.
// Weights are passed re-laid out once at fold time as Wt[cin/8][tap][8][cout] (k-major: k = tap * 8 + ci within a
// stage, output channels contiguous), rows past cin zero, so a stage's weight rows are DMA'd as they lie.
// MFMA: v_mfma_f32_16x16x4_f32, exact fp32 multiply-adds; four k per instruction = four input channels of one tap.
// Wave tile 32 (M) x 64 (N): 2 x 4 blocks, 8 independent accumulators; four waves per workgroup as WMW x WNW.
// Three-stage LDS ring, two stages in flight, one barrier per stage; the four multiplying waves are fed by four
// loader waves that only issue the DMAs (a wave with LDS-DMAs in flight gets an s_waitcnt vmcnt(0) from hipcc in
// front of its LDS reads -- it cannot tell the pending LDS write from the buffer being read -- so a wave that did
// both would wait for the stage it has just requested).
#include "nw_internal.h"
#include <cstdlib>

namespace nw {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __attribute__((aligned(16))) float nw_c3_zeros[4];   // source of the DMA pieces outside the plane
constexpr int C3K = 8;      // input channels per stage
constexpr int C3NBUF = 3;

template <int WMW, int WNW, bool VEC>
struct C3Cfg {   // (the K-split form is WMW = WNW = 1: one 32 x 64 tile, the four multiplying waves share its K range)
    static constexpr int TM = 32 * WMW, TN = 64 * WNW;
    static constexpr int A_F = 9 * C3K * TM;                 // floats of weights per stage
    static constexpr int A_DMA = (A_F * 4 + 1023) / 1024;    // 1 KB DMA instructions for them
    static constexpr int A_PER = (A_DMA + 3) / 4;            // ... per wave (the surplus repeats the last piece)
};

// `lrow`: floats per channel row of the span in LDS (a whole number of DMA instructions + 32), `nb`: DMA instructions
// per channel row.  LDS coordinate of pixel p = (y, x): L(p) = y pitch + x; the span starts at L(p0) - pitch - gap.
// KS (small planes: a 32 x 256 tile per image would leave most CUs idle): the workgroup computes ONE 32 x 64 tile, its
// four multiplying waves take the 18 (tap, channel group) pairs of a stage in turn and their partial tiles are added
// through LDS after the loop (fixed order: deterministic).
template <int WMW, int WNW, bool VEC, bool KS = false>
__global__ __launch_bounds__(512, 2) void nw_conv3x3_kernel(
    const float* __restrict__ x, int64_t x_bs, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ res, int64_t res_bs, int post_relu, float* __restrict__ out, int64_t out_bs,
    float* __restrict__ part, int n_img, int cin, int cout, int H, int W, int tiles_per_img, int lrow, int nb,
    int st_chunk) {
    using Cfg = C3Cfg<WMW, WNW, VEC>;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, A_F = Cfg::A_F, A_DMA = Cfg::A_DMA, A_PER = Cfg::A_PER;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int stage_f = A_F + C3K * lrow;                      // floats per stage
    float* ring = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;   // index within the role: consumer waves 0-3, loader waves 4-7
    const int img = blockIdx.x / tiles_per_img, p0 = (blockIdx.x - img * tiles_per_img) * TN;
    const int m0 = blockIdx.y * TM;
    const int HW = H * W;
    const int wm = KS ? 0 : 32 * (wave / WNW), wn = KS ? 0 : 64 * (wave % WNW);
    // K may be split over blockIdx.z (KS form only: small planes with many input channels): this workgroup's stages
    const int nst_all = (cin + C3K - 1) / C3K;
    const int s_lo = KS ? blockIdx.z * st_chunk : 0;
    const int nst = KS ? min(nst_all - s_lo, st_chunk) : nst_all;
    const float* ximg = x + (int64_t)img * x_bs;

    // ---- DMA plans.  Weights: the stage's 9*8 rows of TM floats; when cout == TM they are one contiguous block,
    // otherwise rows of TM*4 bytes at a stride of cout*4.  Piece t of 1 KB: rows (1024 / (4 TM)) t ...
    constexpr int AROWS = 256 / TM;                 // weight rows per 1 KB instruction (8, 4 or 2)
    constexpr int AL = TM / 4;                      // lanes per row
    const char* asrc[A_PER];
    int adst[A_PER];
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
        int t = wave + 4 * u;
        if (t > A_DMA - 1) t = A_DMA - 1;           // surplus slots repeat the last piece (same bytes, same place)
        const int row = AROWS * t + lane / AL;      // k-row of the stage: tap * 8 + ci
        asrc[u] = reinterpret_cast<const char*>(wt + (int64_t)row * cout + m0 + 4 * (lane % AL));
        adst[u] = 256 * t;                          // floats from the stage's start
    }
    // Span: channel row c of the stage gets nb instructions; instruction b covers span elements [64 b, 64 b + 64)
    // (float4 groups when VEC).  Per wave: 2 channels (8 per stage / 4 waves).
    constexpr int per_el = VEC ? 4 : 1, GAP = VEC ? 4 : 1;
    const int pitch = W + GAP;
    const int y0 = p0 / W;
    const int origin = y0 * pitch + (p0 - y0 * W) - pitch - GAP;   // LDS coordinate of the span's first element (VEC: % 4 == 0)
    // source of this lane's piece of instruction b, as an offset into the plane; -1: not a pixel (gap, outside rows)
    int soff[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int lc = origin + (64 * b + lane) * per_el;
        const int r = (lc + pitch) / pitch - 1;     // floor(lc / pitch) for lc >= -pitch (the origin is at least that)
        const int c = lc - r * pitch;
        soff[b] = (r < 0 || r >= H || c >= W) ? -1 : r * W + c;
    }
    auto issue = [&](int s) {
        float* st = ring + (unsigned)(s % C3NBUF) * stage_f;
        const int64_t koff = (int64_t)(s_lo + s) * 9 * C3K * cout * 4;    // bytes: the stage's first weight row
#pragma unroll
        for (int u = 0; u < A_PER; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[u] + koff),
                                             (__attribute__((address_space(3))) void*)(st + adst[u]), 16, 0, 0);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int cl = 2 * wave + cc;           // channel row of the stage
            const int ci = min((s_lo + s) * C3K + cl, cin - 1);   // rows past cin meet zero weights: any finite data will do
            const float* plane = ximg + (int64_t)ci * HW;
            float* drow = st + A_F + cl * lrow + 16 * (cl & 1);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b >= nb) break;
                const float* src = soff[b] < 0 ? nw_c3_zeros : plane + soff[b];
                if (VEC)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(drow + 256 * b), 16, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(drow + 64 * b), 4, 0, 0);
            }
        }
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a][e] = f32x4{0.f, 0.f, 0.f, 0.f};

    // two stages in flight; vmcnt cannot take a run-time count, so a stage is waited for by draining the queue
    // down to one stage's worth with a loop over the (at most two) outstanding ones
    const int per = A_PER + 2 * nb;                 // DMAs per loader wave and stage (wave-uniform)
    auto wait_all_but = [&](int keep) {             // keep = 0 or `per`
        if (keep == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else {
            // per <= 5 + 2 * 8: enumerate
            switch (per) {
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
                case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
                case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
                case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
                case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
    };
    if (wave8 >= 4) {
        issue(0);
        if (nst > 1) issue(1);
        wait_all_but(nst > 1 ? per : 0);
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nst; ++s) {
            if (s + 2 < nst) issue(s + 2);              // into the buffer the consumers left at the last barrier
            wait_all_but(s + 2 < nst ? per : 0);        // stage s + 1 has landed (this wave's share)
            __builtin_amdgcn_s_barrier();
        }
        if (KS) __builtin_amdgcn_s_barrier();           // (the consumers' reduction barrier)
        return;
    }
    __builtin_amdgcn_s_barrier();                   // stage 0 has landed

    // B fragment of tap t, channel group kk (channels 4 kk + g of the stage), N-block e: row (4 kk + g), element
    //   L(p) - origin + (ky - 1) pitch + (kx - 1),  p = p0 + wn + 16 e + i  (columns past the plane repeat its last pixel)
    int bidx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int p = min(p0 + wn + 16 * e + i, HW - 1);
        const int yy = p / W;
        bidx[e] = yy * pitch + (p - yy * W) - origin - pitch - 1;   // + ky pitch + kx
    }
    for (int s = 0; s < nst; ++s) {
        const float* As = ring + (unsigned)(s % C3NBUF) * stage_f;
        const float* Bs = As + A_F;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = (t / 3) * pitch + (t % 3);
#pragma unroll
            for (int kk = 0; kk < C3K / 4; ++kk) {
                if (KS && ((2 * t + kk) & 3) != wave) continue;   // this pair belongs to another wave
                const int cl = 4 * kk + g;
                const float* brow = Bs + cl * lrow + 16 * (cl & 1) + toff;
                float bv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = brow[bidx[e]];
                const float2 a2 = *reinterpret_cast<const float2*>(As + (t * C3K + cl) * TM + wm + 2 * i);
                const float av[2] = {a2.x, a2.y};
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[a][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[e], acc[a][e], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- store: acc[a][e][r] of lane (i, g) = out[m0 + wm + 2 (4 g + r) + a][p0 + wn + 16 e + i]
    const float lo = post_relu ? 0.f : -INFINITY;
    float* oimg = out + (int64_t)img * out_bs;
    const float* rimg = res ? res + (int64_t)img * res_bs : nullptr;
    if (KS) {
        // the four partial tiles meet in LDS (the ring is free: the loop's last barrier is behind every wave); wave w
        // then owns N-block e = w: value [e][r][a] of lane l from wave v sits at ((v * 4 + e) * 8 + 2 r + a) * 64 + l
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 2; ++a) ring[((wave * 4 + e) * 8 + 2 * r + a) * 64 + lane] = acc[a][e][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += ring[((w * 4 + wave) * 8 + 2 * r + a) * 64 + lane];
                acc[a][0][r] = v;
            }
    }
#pragma unroll
    for (int e0 = 0; e0 < 4; ++e0) {
        if (KS && e0 > 0) break;
        const int e = KS ? wave : e0;
        const int p = p0 + wn + 16 * e + i;
        if (p >= HW) continue;
        if (KS && gridDim.z > 1) {   // raw partial sums [z][img][m][p]; nw_conv3x3_reduce_kernel finishes
            float* pz = part + (((int64_t)blockIdx.z * n_img + img) * cout) * HW;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 2; ++a) pz[(int64_t)(m0 + 2 * (4 * g + r) + a) * HW + p] = acc[a][0][r];
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int m = m0 + wm + 2 * (4 * g + r) + a;
                if (m >= cout) continue;
                float v = acc[a][KS ? 0 : e0][r] + (bias ? bias[m] : 0.f);
                if (rimg) v += rimg[(int64_t)m * HW + p];
                oimg[(int64_t)m * HW + p] = fmaxf(v, lo);
            }
    }
}

// out[n, m, p] = post(bias[m] + sum_z part[z][n][m][p] [+ res]), in z order (deterministic)
__global__ __launch_bounds__(256) void nw_conv3x3_reduce_kernel(const float* __restrict__ part, int nz,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ res, int64_t res_bs,
                                                                 int post_relu, float* __restrict__ out, int64_t out_bs,
                                                                 int n_img, int cout, int HW) {
    const int64_t per_img = (int64_t)cout * HW, total = per_img * n_img;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int img = (int)(idx / per_img);
    const int64_t rem = idx - (int64_t)img * per_img;
    const int m = (int)(rem / HW);
    float v = part[idx];
    for (int z = 1; z < nz; ++z) v += part[(int64_t)z * total + idx];
    if (bias) v += bias[m];
    if (res) v += res[(int64_t)img * res_bs + rem];
    out[(int64_t)img * out_bs + rem] = post_relu ? fmaxf(v, 0.f) : v;
}

}  // namespace
}  // namespace nw

// span geometry: rows `pitch` = W + gap apart in LDS; a tile of tn pixels starts at most W - 1 into a row
static void conv3x3_span(int tn, int W, bool vec, int* lrow, int* nb) {
    const int gap = vec ? 4 : 1, pitch = W + gap;
    const int rows = (tn + W - 2) / W + 1;                      // image rows a tile can touch
    const int L = rows * pitch + 2 * pitch + 2 * gap;          // + the row above and below, + the corners
    const int per = vec ? 256 : 64;                            // floats per DMA instruction
    *nb = (L + per - 1) / per;
    *lrow = *nb * per + 32;                                    // + the 16-float stagger of odd rows
}

// Tile form for a shape.  32 x 256 for narrow outputs on big planes, 64 x 128 otherwise, 128 x 64 for small planes; when
// that leaves fewer than three quarters of the CUs with a workgroup: 32 x 64 tiles whose K range is shared by the
// workgroup's four multiplying waves (ks), and if even those are too few and there are enough input channels, the K
// range is also split over `zs` workgroups whose partial tiles a second kernel adds in order.
struct Conv3x3Plan {
    int wmw, wnw, zs, st_chunk;
    bool ks;
    int64_t workgroups;
};
static Conv3x3Plan conv3x3_plan(int64_t n, int64_t cin, int64_t cout, int64_t HW) {
    Conv3x3Plan p;
    if (cout % 128 == 0 && HW <= 64) { p.wmw = 4; p.wnw = 1; }
    else if (cout % 64 == 0) { p.wmw = 2; p.wnw = 2; }
    else { p.wmw = 1; p.wnw = 4; }
    p.zs = 1;
    const int64_t nst = (cin + nw::C3K - 1) / nw::C3K;
    p.st_chunk = (int)nst;
    p.workgroups = ((HW + 64 * p.wnw - 1) / (64 * p.wnw)) * n * (cout / (32 * p.wmw));
    p.ks = p.workgroups < 192;
    if (p.ks) {
        p.wmw = p.wnw = 1;
        p.workgroups = ((HW + 63) / 64) * n * (cout / 32);
        if (p.workgroups < 192 && nst >= 8) {
            int64_t zs = (255 + p.workgroups) / p.workgroups;      // towards one workgroup per CU
            if (zs > 8) zs = 8;
            if (zs > nst / 4) zs = nst / 4;                         // at least four stages each
            const int64_t chunk = (nst + zs - 1) / zs;
            p.st_chunk = (int)chunk;
            p.zs = (int)((nst + chunk - 1) / chunk);
            p.workgroups *= p.zs;
        }
    }
    return p;
}

extern "C" size_t nw_conv3x3_workspace_bytes(int64_t n, int64_t cin, int64_t cout, int64_t H, int64_t W) {
    if (n <= 0 || cin <= 0 || cout <= 0 || H <= 0 || W <= 0 || cout % 32 != 0) return 0;
    const Conv3x3Plan p = conv3x3_plan(n, cin, cout, H * W);
    return p.zs > 1 ? (size_t)p.zs * n * cout * H * W * sizeof(float) : 0;
}

// workgroups nw_conv3x3_f32 launches for this shape (callers keep MIOpen for shapes that leave the chip idle)
extern "C" int64_t nw_conv3x3_workgroups(int64_t n, int64_t cin, int64_t cout, int64_t H, int64_t W) {
    if (n <= 0 || cin <= 0 || cout <= 0 || H <= 0 || W <= 0 || cout % 32 != 0) return 0;
    return conv3x3_plan(n, cin, cout, H * W).workgroups;
}

extern "C" int nw_conv3x3_f32(const float* x, int64_t x_batch_stride, const float* w_t, const float* bias,
                              const float* residual, int64_t res_batch_stride, int post_relu, float* out,
                              int64_t out_batch_stride, void* workspace, size_t workspace_bytes, int64_t n, int64_t cin,
                              int64_t cout, int64_t H, int64_t W, void* stream) {
    using namespace nw;
    if (n < 0 || cin < 0 || cout < 0 || H < 0 || W < 0) return NW_ERR_INVALID_ARG;
    if (n == 0 || cout == 0 || H == 0 || W == 0) return NW_OK;
    if (!x || !w_t || !out || cin == 0) return NW_ERR_INVALID_ARG;
    if (cout % 32 != 0 || (reinterpret_cast<uintptr_t>(w_t) & 15) || (reinterpret_cast<uintptr_t>(x) & 3)) return NW_ERR_UNSUPPORTED;
    const int64_t HW = H * W;
    if (HW > 0x3fffffffLL || cin > 0x7fffffffLL || cout > 0x7fffffffLL || x_batch_stride < cin * HW || out_batch_stride < cout * HW ||
        (residual && res_batch_stride < cout * HW))
        return NW_ERR_INVALID_ARG;
    if (HW < 4) return NW_ERR_UNSUPPORTED;
    const bool vec = W % 4 == 0 && x_batch_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const Conv3x3Plan pl = conv3x3_plan(n, cin, cout, HW);
    if (pl.zs > 1 && (!workspace || workspace_bytes < nw_conv3x3_workspace_bytes(n, cin, cout, H, W) ||
                      (reinterpret_cast<uintptr_t>(workspace) & 15)))
        return NW_ERR_WORKSPACE;
    const int tm = 32 * pl.wmw, tn = 64 * pl.wnw;
    const int tiles = (int)((HW + tn - 1) / tn);
    const int64_t gx = (int64_t)tiles * n;
    if (gx > 0x7fffffffLL || cout / tm > 65535 || n * cout * HW > 0x7fffffffffLL) return NW_ERR_INVALID_ARG;
    int lrow, nb;
    conv3x3_span(tn, (int)W, vec, &lrow, &nb);
    if (nb > 8) return NW_ERR_UNSUPPORTED;                       // very wide images: not a backbone shape
    size_t lds = (size_t)C3NBUF * (9 * C3K * tm + C3K * lrow) * sizeof(float);
    if (pl.ks && lds < 32 * 1024) lds = 32 * 1024;               // the reduction of the four partial tiles
    if (lds > 160 * 1024) return NW_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)gx, (unsigned)(cout / tm), (unsigned)pl.zs);
    float* part = static_cast<float*>(workspace);
#define NW_C3(MW_, NW_, V_, KS_)                                                                                         \
    hipLaunchKernelGGL((nw_conv3x3_kernel<MW_, NW_, V_, KS_>), grid, dim3(512), lds, st, x, x_batch_stride, w_t, bias, residual, \
                       res_batch_stride, post_relu, out, out_batch_stride, part, (int)n, (int)cin, (int)cout, (int)H, (int)W, \
                       tiles, lrow, nb, pl.st_chunk)
    if (pl.ks) { if (vec) NW_C3(1, 1, true, true); else NW_C3(1, 1, false, true); }
    else if (pl.wmw == 1) { if (vec) NW_C3(1, 4, true, false); else NW_C3(1, 4, false, false); }
    else if (pl.wmw == 2) { if (vec) NW_C3(2, 2, true, false); else NW_C3(2, 2, false, false); }
    else { if (vec) NW_C3(4, 1, true, false); else NW_C3(4, 1, false, false); }
#undef NW_C3
    if (pl.zs > 1) {
        const int64_t total = n * cout * HW;
        hipLaunchKernelGGL(nw_conv3x3_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, pl.zs, bias,
                           residual, res_batch_stride, post_relu, out, out_batch_stride, (int)n, (int)cout, (int)HW);
    }
    NW_CHECK_LAUNCH();
    return NW_OK;
}
