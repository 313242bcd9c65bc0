// conv1x1.hip -- 1x1 convolution of the backbones' inference path as ONE matrix-core kernel with the
// BatchNorm -> ReLU in front of it and the (folded BatchNorm) bias + ReLU behind it fused in (gfx950 / MI355X only).
//
// Replaces, in the folded inference copy of DenseNet (model/densenet.py:33-60 `_DenseLayer`: norm1 - relu1 - conv1 (1x1)
// - norm2 - relu2; :82-91 `_Transition`: norm - relu - conv (1x1)) and of CIFAR_DenseNet (model/densenet3.py:10-22), the
// sequence  scale-shift-ReLU pass -> GEMM (Tensile) -> bias add -> ReLU pass:  per forward of DenseNet-121 over 64
// images @224 that was 61 GEMM launches (2.55 ms) + 181 elementwise launches (2.55 ms) of an 8.4 ms forward.
//
//     out[n, co, p] = post( bias[co] + sum_ci W[co, ci] * pre(x[n, ci, p]) ),   pre(v) = max(a_ci v + b_ci, 0)
//
// As a GEMM per image: M = cout, N = pixels (the hw plane is contiguous in NCHW: the B operand is row-major as it
// lies in memory, a channel PREFIX of a wider dense-block slab included), K = cin.  The pixel axis runs over all
// images (column = n * hw + p), so small planes (14x14, 7x7) still fill 128-column tiles.
// Arithmetic: v_mfma_f32_16x16x4_f32 -- exact fp32 multiply-adds, the reference's precision, on the matrix cores
// (157 TFLOP/s peak; the prologue/epilogue passes it absorbs were HBM-bound).
//   tile 128 (M) x TN (N = 128 or 64) per 256-thread workgroup, four waves of 64 x TN/2, K in steps of 16 through
//   two LDS stages ([k][m] / [k][n] rows, stride TN+16 / 144 floats: 16 mod 32 banks), next step's global loads in
//   flight under the MFMAs, one barrier per step.  The weights are passed TRANSPOSED ([cin][cout], made once when the
//   inference copy is folded) so that both operands are staged with 16-byte accesses.
#include "nw_internal.h"
#include <cstdlib>

namespace nw {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CM = 128, CK = 16, CLDA = 144;

template <int TN, bool VEC>
__global__ __launch_bounds__(256, 2) void nw_conv1x1_kernel(
    const float* __restrict__ x, int64_t x_bs, const float* __restrict__ pre_a, const float* __restrict__ pre_b,
    int pre_relu, const float* __restrict__ wt, const float* __restrict__ bias, int post_relu,
    float* __restrict__ out, int64_t out_bs, int n_img, int cin, int cout, int hw) {
    constexpr int LDB = TN + 16;      // floats per k-row of the B stage: 16 mod 32 banks
    constexpr int NB = TN / 32;       // 16-column blocks per wave
    constexpr int BCH = TN / 4;       // float4 chunks per k-row of B
    constexpr int BIT = CK * BCH / 256;  // B chunks per thread and step (2 for TN = 128, 1 for TN = 64)
    __shared__ __attribute__((aligned(16))) float As[2][CK * CLDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][CK * LDB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * CM;
    const int64_t col0 = (int64_t)blockIdx.x * TN, ncols = (int64_t)n_img * hw;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * (TN / 2);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[4][NB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's B chunks: k-row (tid / BCH + (256 / BCH) * u), columns 4 * (tid % BCH) .. +3 of the tile
    const int bk = tid / BCH, bc = 4 * (tid % BCH);
    const int64_t gcol = col0 + bc;
    // column -> (image, pixel); VEC: hw % 4 == 0, so a float4 never straddles two images
    const float* xcol[4];
    bool colok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t c = gcol + (VEC ? 0 : e);
        colok[e] = c < ncols;
        const int64_t cc = colok[e] ? c : 0;
        const int64_t img = cc / hw, p = cc - img * hw;
        xcol[e] = x + img * x_bs + p;
    }
    float4 ra[2], rb[BIT];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // A: Wt[k][m], 16 k-rows x 128 m = 512 float4, 2 per thread
            const int k = k0 + (tid >> 5) + 8 * u, m = m0 + 4 * (tid & 31);
            ra[u] = (k < cin && m < cout) ? *reinterpret_cast<const float4*>(wt + (int64_t)k * cout + m) : zero4;
        }
#pragma unroll
        for (int u = 0; u < BIT; ++u) {
            const int k = k0 + bk + (256 / BCH) * u;
            float4 v = zero4;
            if (k < cin) {
                if (VEC) {
                    if (colok[0]) v = *reinterpret_cast<const float4*>(xcol[0] + (int64_t)k * hw);
                } else {
                    if (colok[0]) v.x = xcol[0][(int64_t)k * hw];
                    if (colok[1]) v.y = xcol[1][(int64_t)k * hw];
                    if (colok[2]) v.z = xcol[2][(int64_t)k * hw];
                    if (colok[3]) v.w = xcol[3][(int64_t)k * hw];
                }
                if (pre_a) {  // eval-mode BatchNorm of input channel k, then ReLU
                    const float a = pre_a[k], b = pre_b[k];
                    v = make_float4(__builtin_fmaf(a, v.x, b), __builtin_fmaf(a, v.y, b), __builtin_fmaf(a, v.z, b), __builtin_fmaf(a, v.w, b));
                }
                if (pre_relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                // columns past the end feed output columns that are never stored; rows past cin must be exact zeros
            }
            rb[u] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            *reinterpret_cast<float4*>(&As[buf][((tid >> 5) + 8 * u) * CLDA + 4 * (tid & 31)]) = ra[u];
#pragma unroll
        for (int u = 0; u < BIT; ++u)
            *reinterpret_cast<float4*>(&Bs[buf][(bk + (256 / BCH) * u) * LDB + bc]) = rb[u];
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < CK / 4; ++kk) {
            const int kr = 4 * kk + (lane >> 4);
            float a[4], b[NB];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[buf][kr * CLDA + wm + 16 * i + (lane & 15)];
#pragma unroll
            for (int j = 0; j < NB; ++j) b[j] = Bs[buf][kr * LDB + wn + 16 * j + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nsteps = (cin + CK - 1) / CK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
        const bool more = t + 1 < nsteps;
        if (more) gload((t + 1) * CK);
        compute(t & 1);
        if (more) sstore((t + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane (i = lane & 15, g = lane >> 4) holds rows wm + 16 bi + 4 g + r, column wn + 16 bj + i
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int64_t c = col0 + wn + 16 * j + (lane & 15);
        if (c >= ncols) continue;
        const int64_t img = c / hw, p = c - img * hw;
        float* ocol = out + img * out_bs + p;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * i + 4 * (lane >> 4) + r;
                if (m >= cout) continue;
                float v = acc[i][j][r] + (bias ? bias[m] : 0.f);
                if (post_relu) v = fmaxf(v, 0.f);
                ocol[(int64_t)m * hw] = v;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The main kernel (cout % 128 == 0, hw % 4 == 0): LDS-DMA pipeline, no register staging.
//   * both operand tiles go HBM/L2 -> LDS by global_load_lds_dwordx4 into a four-stage ring ([k][128] weight rows,
//     [k][TN] activation rows, 16 k per stage), three stages ahead, counted vmcnt + one raw barrier per stage.
//     Waves 0-3 multiply, waves 4-7 only issue the DMAs (4 per wave and stage): a wave with LDS-DMAs in flight gets
//     an s_waitcnt vmcnt(0) from hipcc in front of its LDS reads (it cannot tell the pending LDS write from the
//     buffer being read), i.e. the ring would drain at every stage.  Two workgroups per CU.
//   * fragments by ds_read_b128: lane (i, g) takes FOUR consecutive rows (columns) of k-row 4 kk + g and uses them
//     as the operand of four MFMA blocks -- the block's row (column) index i then stands for row 4 i + e; the
//     output tile is read back through the same map, which gives every lane four consecutive pixels per output
//     channel: 16-byte stores, 256 contiguous bytes per 16 lanes.  Two LDS reads per 16 MFMAs, conflict-free on
//     unpadded rows (a 16-lane service group of ds_read_b128 covers one 256-byte bank row).
//   * the BatchNorm + ReLU in front of the convolution is applied to the activation fragment after the LDS read
//     (8 VALU ops per 16 MFMAs; per-channel factors sit in LDS), bias + ReLU behind it in the store.
//   * K may be split over blockIdx.z (small planes: 14x14 and 7x7 leave 98 / 25 column tiles for 256 CUs); partial
//     tiles go to a workspace and nw_conv1x1_reduce_kernel adds them in order.
// ---------------------------------------------------------------------------------------------------------------
constexpr int DK = 16, DNBUF = 4;

// B4: planes whose size is not a multiple of 4 (7x7): the activation rows go by 4-byte DMAs, one column per lane
// (a 16-byte piece could straddle two images), and the results are stored one by one.  TN = 64 only.
template <int TN, bool B4 = false>
__global__ __launch_bounds__(512, 2) void nw_conv1x1_dma_kernel(
    const float* __restrict__ x, int64_t x_bs, const float* __restrict__ pre_a, const float* __restrict__ pre_b,
    int pre_relu, const float* __restrict__ wt, const float* __restrict__ bias, int post_relu,
    float* __restrict__ out, int64_t out_bs, float* __restrict__ part, int n_img, int cin, int cout, int hw,
    int k_chunk) {
    constexpr int WM = (TN == 128) ? 64 : 32;    // rows per wave: 2 x 2 waves of 64 x 64, or 4 x 1 waves of 32 x 64
    constexpr int NA = WM / 16, NB = 4;          // MFMA blocks per wave along M and N
    constexpr int A_F4 = DK * 128 / 4, B_F4 = DK * TN / 4, ST_F4 = A_F4 + B_F4;   // float4 per stage
    constexpr int NDA = 2, NDB = B4 ? 4 : TN / 64;   // DMA instructions per wave and stage: weights, activations
    static_assert(!B4 || TN == 64, "4-byte activation DMAs: one 64-column k-row per instruction");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* ring = reinterpret_cast<float4*>(smem);
    float2* pre = reinterpret_cast<float2*>(smem + (size_t)DNBUF * ST_F4 * 16);   // [k_chunk] (a, b) of channel kb + k
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;   // index within the role (consumer 0-3 / loader 0-3)
    const int m0 = blockIdx.y * 128;
    const int64_t col0 = (int64_t)blockIdx.x * TN, ncols = (int64_t)n_img * hw;
    const int kb = blockIdx.z * k_chunk, ke = min(cin, kb + k_chunk);
    const int nst = (ke - kb + DK - 1) / DK;
    const int wm = (TN == 128) ? (wave >> 1) * 64 : wave * 32, wn = (TN == 128) ? (wave & 1) * 64 : 0;

    for (int k = tid; k < nst * DK; k += 512) {
        const int kc = min(kb + k, cin - 1);
        pre[k] = pre_a ? make_float2(pre_a[kc], pre_b[kc]) : make_float2(1.f, 0.f);
    }
    const float lo = pre_relu ? 0.f : -INFINITY;

    // ---- DMA sources.  Weights: instruction t of the stage covers k-rows 2t, 2t+1 (512 bytes each); this wave
    // issues t = 2 wave, 2 wave + 1.  Activations: an instruction covers 1024 / (4 TN) k-rows.
    const char* asrc[NDA];
#pragma unroll
    for (int u = 0; u < NDA; ++u) {
        const int t = NDA * wave + u;
        asrc[u] = reinterpret_cast<const char*>(wt + (int64_t)(kb + 2 * t + (lane >> 5)) * cout + m0 + 4 * (lane & 31));
    }
    constexpr int BROWS = B4 ? 1 : 256 / TN;     // k-rows per activation DMA (2 or 4; one with 4-byte pieces)
    constexpr int BL = B4 ? 64 : TN / 4;         // lanes per k-row
    const char* bsrc[NDB];
    int brow[NDB];
    {
        int64_t c = col0 + (B4 ? 1 : 4) * (lane % BL);
        if (c > ncols - (B4 ? 1 : 4)) c = ncols - (B4 ? 1 : 4);   // columns past the end repeat the last ones (never stored)
        const int64_t img = c / hw, p = c - img * hw;
#pragma unroll
        for (int u = 0; u < NDB; ++u) {
            brow[u] = BROWS * (NDB * wave + u) + lane / BL;
            bsrc[u] = reinterpret_cast<const char*>(x + img * x_bs + p);
        }
    }
    auto issue = [&](int s) {
        float4* st = ring + (unsigned)(s % DNBUF) * ST_F4;
        const int k0 = s * DK;
#pragma unroll
        for (int u = 0; u < NDA; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[u] + (int64_t)k0 * cout * 4),
                                             (__attribute__((address_space(3))) void*)(st + 64 * (NDA * wave + u)), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < NDB; ++u) {
            const int k = min(kb + k0 + brow[u], cin - 1);   // rows past cin meet zero weight rows: any finite data will do
            const char* src = bsrc[u] + (int64_t)k * hw * 4;
            if (B4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(reinterpret_cast<float*>(st + A_F4) + 64 * (NDB * wave + u)),
                                                 4, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(st + A_F4 + 64 * (NDB * wave + u)), 16, 0, 0);
        }
    };
    constexpr int PER = NDA + NDB;               // DMAs per wave and stage
    f32x4 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wave8 >= 4) {
        // loader waves.  prologue: three stages in flight, the first one landed
#pragma unroll
        for (int s = 0; s < DNBUF - 1; ++s)
            if (s < nst) issue(s);
        if (nst >= DNBUF - 1) wait_vmcnt<2 * PER>();   // (tile_dma.h: the count is checked against the 6-bit field at compile time)
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's share of `pre`)
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nst; ++s) {
            if (s + DNBUF - 1 < nst) issue(s + DNBUF - 1);   // into the buffer the consumers left at the last barrier
            // stage s + 1 of this wave has landed once at most the two youngest stages are in flight
            if (s + DNBUF - 1 < nst) wait_vmcnt<2 * PER>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // stage 0 has landed, `pre` is published

    for (int s = 0; s < nst; ++s) {
        const float4* As = ring + (unsigned)(s % DNBUF) * ST_F4;
        const float4* Bs = As + A_F4;
#pragma unroll
        for (int kk = 0; kk < DK / 4; ++kk) {
            const int kr = 4 * kk + g;
            const float2 ab = pre[s * DK + kr];
            float4 b4 = Bs[kr * (TN / 4) + (wn >> 2) + i];
            b4 = make_float4(fmaxf(__builtin_fmaf(ab.x, b4.x, ab.y), lo), fmaxf(__builtin_fmaf(ab.x, b4.y, ab.y), lo),
                             fmaxf(__builtin_fmaf(ab.x, b4.z, ab.y), lo), fmaxf(__builtin_fmaf(ab.x, b4.w, ab.y), lo));
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
            float av[NA];
            if (NA == 4) {
                const float4 a4 = As[kr * 32 + (wm >> 2) + i];
                av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[NA - 1] = a4.w;
            } else {
                const float2 a2 = reinterpret_cast<const float2*>(As)[kr * 64 + (wm >> 1) + i];
                av[0] = a2.x; av[NA - 1] = a2.y;
            }
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- store.  acc[a][b][r] of lane (i, g) = out[m0 + wm + NA (4 g + r) + a][col0 + wn + 4 i + b]
    const int64_t c = col0 + wn + 4 * i;
    if (c >= ncols) return;
    const bool final_ = gridDim.z == 1;
    const int64_t ncols4 = (ncols + 3) & ~(int64_t)3;   // row stride of the partial tiles
    const float lo2 = post_relu ? 0.f : -INFINITY;
    if (!final_) {
        float* pbase = part + ((int64_t)blockIdx.z * cout) * ncols4 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int m = m0 + wm + NA * (4 * g + r) + a;
                *reinterpret_cast<float4*>(pbase + (int64_t)m * ncols4) = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
            }
        return;
    }
    if (!B4) {
        const int64_t img = c / hw, p = c - img * hw;
        float* obase = out + img * out_bs + p;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int m = m0 + wm + NA * (4 * g + r) + a;
                const float bs = bias ? bias[m] : 0.f;
                *reinterpret_cast<float4*>(obase + (int64_t)m * hw) =
                    make_float4(fmaxf(acc[a][0][r] + bs, lo2), fmaxf(acc[a][1][r] + bs, lo2), fmaxf(acc[a][2][r] + bs, lo2),
                                fmaxf(acc[a][3][r] + bs, lo2));
            }
    } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (c + b >= ncols) continue;
            const int64_t img = (c + b) / hw, p = (c + b) - img * hw;
            float* ocol = out + img * out_bs + p;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int m = m0 + wm + NA * (4 * g + r) + a;
                    ocol[(int64_t)m * hw] = fmaxf(acc[a][b][r] + (bias ? bias[m] : 0.f), lo2);
                }
        }
    }
}

// out[n, m, p] = post(bias[m] + sum_z part[z][m][col]), col = n hw + p  (4 columns per thread; rows of `part` are
// ncols rounded up to 4 floats apart)
__global__ __launch_bounds__(256) void nw_conv1x1_reduce_kernel(const float* __restrict__ part, int nz,
                                                                 const float* __restrict__ bias, int post_relu,
                                                                 float* __restrict__ out, int64_t out_bs, int cout,
                                                                 int hw, int64_t ncols) {
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int m = blockIdx.y;
    if (c >= ncols) return;
    const int64_t ncols4 = (ncols + 3) & ~(int64_t)3;
    float4 a = *reinterpret_cast<const float4*>(part + (int64_t)m * ncols4 + c);
    for (int z = 1; z < nz; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)z * cout + m) * ncols4 + c);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const float bs = bias ? bias[m] : 0.f, lo = post_relu ? 0.f : -INFINITY;
    a = make_float4(fmaxf(a.x + bs, lo), fmaxf(a.y + bs, lo), fmaxf(a.z + bs, lo), fmaxf(a.w + bs, lo));
    if ((hw & 3) == 0 && (out_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const int64_t img = c / hw, p = c - img * hw;
        *reinterpret_cast<float4*>(out + img * out_bs + (int64_t)m * hw + p) = a;
    } else {
        const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c + e >= ncols) break;
            const int64_t img = (c + e) / hw, p = (c + e) - img * hw;
            out[img * out_bs + (int64_t)m * hw + p] = v[e];
        }
    }
}

}  // namespace
}  // namespace nw

// K split of the DMA kernel: enough workgroups for two per CU, at least 64 of K per chunk
static int conv1x1_ksplit(int64_t tiles, int64_t cin) {
    int64_t z = tiles >= 512 ? 1 : (512 + tiles - 1) / tiles;
    const int64_t maxz = cin / 64 > 1 ? cin / 64 : 1;
    if (z > maxz) z = maxz;
    if (z > 16) z = 16;
    return (int)z;
}
static bool conv1x1_dma_ok(const float* x, int64_t x_bs, const float* w_t, int64_t cout, int64_t hw, int64_t ncols) {
    (void)x_bs; (void)hw;
    return cout % 128 == 0 && ncols >= 4 && (reinterpret_cast<uintptr_t>(w_t) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 3) == 0;
}
// 16-byte activation DMAs and stores need whole 4-pixel groups inside one image, 16-byte aligned
static bool conv1x1_vec16(const float* x, int64_t x_bs, const float* out, int64_t out_bs, int64_t hw) {
    return hw % 4 == 0 && x_bs % 4 == 0 && out_bs % 4 == 0 &&
           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
}
static int conv1x1_tn(int64_t ncols, int64_t cout, bool vec16) {
    return (vec16 && ((ncols + 127) / 128) * (cout / 128) >= 512) ? 128 : 64;
}

extern "C" size_t nw_conv1x1_workspace_bytes(int64_t n, int64_t cin, int64_t cout, int64_t hw) {
    if (n <= 0 || cin <= 0 || cout <= 0 || hw <= 0 || cout % 128 != 0 || n * hw < 4) return 0;
    const int64_t ncols = n * hw;
    const int tn = conv1x1_tn(ncols, cout, hw % 4 == 0);
    const int z = conv1x1_ksplit(((ncols + tn - 1) / tn) * (cout / 128), cin);
    return z > 1 ? (size_t)z * cout * ((ncols + 3) & ~(int64_t)3) * sizeof(float) : 0;
}

extern "C" int nw_conv1x1_f32(const float* x, int64_t x_batch_stride, const float* pre_scale, const float* pre_shift,
                              int pre_relu, const float* w_t, const float* bias, int post_relu, float* out,
                              int64_t out_batch_stride, void* workspace, size_t workspace_bytes, int64_t n, int64_t cin,
                              int64_t cout, int64_t hw, void* stream) {
    using namespace nw;
    if (n < 0 || cin < 0 || cout < 0 || hw < 0) return NW_ERR_INVALID_ARG;
    if (n == 0 || cout == 0 || hw == 0) return NW_OK;
    if (!x || !w_t || !out || (pre_scale && !pre_shift)) return NW_ERR_INVALID_ARG;
    if (cout % 4 != 0 || (reinterpret_cast<uintptr_t>(w_t) & 15)) return NW_ERR_UNSUPPORTED;  // 16-byte weight loads
    if (n * hw > 0x7fffffffLL * 32 || cin > 0x7fffffffLL || cout > 0x7fffffffLL || hw > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    if (x_batch_stride < cin * hw || out_batch_stride < cout * hw) return NW_ERR_INVALID_ARG;
    const int64_t ncols = n * hw;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool no_dma = false;   // (the generic register-staged kernel serves the shapes the LDS-DMA kernel refuses)
    if (cin > 0 && !no_dma && conv1x1_dma_ok(x, x_batch_stride, w_t, cout, hw, ncols)) {
        const bool v16 = conv1x1_vec16(x, x_batch_stride, out, out_batch_stride, hw);
        const int tn = conv1x1_tn(ncols, cout, v16);
        const int64_t gx = (ncols + tn - 1) / tn;
        const int z = conv1x1_ksplit(gx * (cout / 128), cin);
        if (gx > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
        int k_chunk = (int)((cin + z - 1) / z);
        k_chunk = (k_chunk + DK - 1) / DK * DK;
        const int nz = (int)((cin + k_chunk - 1) / k_chunk);
        float* part = nullptr;
        if (nz > 1) {
            const size_t need = (size_t)nz * cout * ((ncols + 3) & ~(int64_t)3) * sizeof(float);
            if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return NW_ERR_WORKSPACE;
            part = static_cast<float*>(workspace);
        }
        const size_t stage = (size_t)(DK * 128 + DK * tn) * 4;
        const size_t lds = DNBUF * stage + (size_t)k_chunk * 8;
        const dim3 grid((unsigned)gx, (unsigned)(cout / 128), (unsigned)nz);
#define NW_C1D(TN_, B4_)                                                                                                  \
    hipLaunchKernelGGL((nw_conv1x1_dma_kernel<TN_, B4_>), grid, dim3(512), lds, st, x, x_batch_stride, pre_scale, pre_shift, \
                       pre_relu, w_t, bias, post_relu, out, out_batch_stride, part, (int)n, (int)cin, (int)cout, (int)hw, k_chunk)
        if (!v16) NW_C1D(64, true);
        else if (tn == 128) NW_C1D(128, false);
        else NW_C1D(64, false);
#undef NW_C1D
        if (nz > 1)
            hipLaunchKernelGGL(nw_conv1x1_reduce_kernel, dim3((unsigned)((ncols / 4 + 256) / 256), (unsigned)cout), dim3(256), 0, st,
                               part, nz, bias, post_relu, out, out_batch_stride, (int)cout, (int)hw, ncols);
        NW_CHECK_LAUNCH();
        return NW_OK;
    }
    const bool vec = hw % 4 == 0 && x_batch_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const unsigned my = (unsigned)((cout + CM - 1) / CM);
    // 128-column tiles while they fill the chip twice over, else 64-column ones
    const bool wide = ((ncols + 127) / 128) * my >= 512;
    const int tn = wide ? 128 : 64;
    const int64_t gx = (ncols + tn - 1) / tn;
    if (gx > 0x7fffffffLL || my > 65535u) return NW_ERR_INVALID_ARG;
#define NW_C1(TN_, V_)                                                                                           \
    hipLaunchKernelGGL((nw_conv1x1_kernel<TN_, V_>), dim3((unsigned)gx, my), dim3(256), 0, st, x, x_batch_stride,  \
                       pre_scale, pre_shift, pre_relu, w_t, bias, post_relu, out, out_batch_stride, (int)n, (int)cin, \
                       (int)cout, (int)hw)
    if (wide) { if (vec) NW_C1(128, true); else NW_C1(128, false); }
    else { if (vec) NW_C1(64, true); else NW_C1(64, false); }
#undef NW_C1
    NW_CHECK_LAUNCH();
    return NW_OK;
}
