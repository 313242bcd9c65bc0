// conv_wgrad.hip -- weight gradient of the backbones' stride-1 convolutions on the fp16 matrix cores at fp32-grade
// accuracy (gfx950 / MI355X only).  Replaces what autograd derives for the weight of F.conv2d at model/densenet.py:33-60
// (conv1 1x1, conv2 3x3), :82-91 (transition 1x1) and model/resnet.py:31-66 (3x3) in loss.backward(), train.py:414:
//
//     dW[co, ky, kx, ci] = sum_{n, y, x} gy[n, y, x, co] * x[n, y + ky - p, x + kx - p, ci]          (stride 1, p = (k - 1) / 2)
//
// A GEMM whose contraction runs over PIXELS: both operands are "k-major" (a pixel is a row of channels), so their LDS
// images are read with the transposing ds_read_b64_tr_b16 (bwd_split.hip's second product).  fp32 NHWC activations are
// split into fp16 pairs on the way into LDS with one power of two per tensor (their amax records), like conv_nhwc.hip.
// The contraction index is a position of a VIRTUAL raster with one zero column behind every image row and one zero row
// behind every image (the loaders write the zeros): the taps of a 3x3 kernel are then constant row shifts of ONE ring of
// x rows in LDS (a 32-position stage adds 32 rows to the ring, every row is fetched once), and no border test exists.
// The position axis is split over workgroups; partial tiles are added in chunk order by a second kernel (deterministic).
//   WIDE  (1x1): 128 output x 128 input channels per workgroup, 2 x 2 waves of 64 x 64.
//   TAPS9 (3x3, Cout % 32 == 0): 32 output x 64 input channels x 9 taps per workgroup; wave w takes input-channel block w.
#include "nw_internal.h"
#include "tile_dma.h"
#include <cstdlib>

namespace nw {
namespace {

typedef __fp16 fp16x4w __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 halfx4w __attribute__((ext_vector_type(4)));
constexpr int WG_SLOTS = 256;

struct WgradP {
    const float* x;
    const float* amax_x;
    const float* pre_x;      // nullable: [3][Cin] = mean | a | beta -- the x operand is relu((x - mean) a + beta), applied by the loaders
                             //   (a convolution whose forward read x that way: conv_nhwc.hip's ConvP::pre); amax_x bounds that tensor
    const float* gy;
    const float* amax_g;
    // ROW RUNS (rr_s > 0; 1x1 plan only): the x operand of output pixel (n, yo, xo) is not a pixel's channels but the 32
    // consecutive floats of a 4-channel NHWC input that one kernel ROW of a few-channel convolution reads there (conv_nhwc.hip's
    // ROWRUN mode: the 7x7 / 2 stem over the zero-padded RGB input): input row rr_s yo + rr_row0, pixels rr_s xo + rr_col0 .. + 7.
    // With "channel" 32 ky + 4 kx + ci the whole dW is ONE 1x1 weight gradient between gy and KH x 32 such channels.
    int rr_s, rr_H, rr_W, rr_row0, rr_col0;
    float* part;             // [ks][Cout][T][Cin]
    const float4* zeros;
    int N, H, W, Cin, Cout, T, KW, pad;
    int ldx, ldg;            // floats between consecutive pixels of x / gy (>= Cin / Cout)
    int IP, IMG;             // virtual raster: row stride (W + 1 with taps, W without), image stride
    int nstage;              // 32-position stages in all
    int ks, spc;             // position chunks, stages per chunk
    int co_tiles, ci_tiles;
};

__device__ __forceinline__ float amax_rec(const float* rec, int lane) {   // max of an amax record, by one wave
    const float4 v = reinterpret_cast<const float4*>(rec)[lane];
    return wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
}
__device__ __forceinline__ int unit_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <bool TAPS9>
struct WgCfg {
    static constexpr int COT = TAPS9 ? 32 : 128, CIT = TAPS9 ? 64 : 128;
    static constexpr int GRS = TAPS9 ? 256 : 512;         // bytes of a gy row in LDS (32 channels use half of a 256-byte row)
    static constexpr int XRS = CIT * 4;                    // bytes of an x row in LDS
    static constexpr int NBG = 3;                          // gy stage buffers
    static constexpr int RING = TAPS9 ? 256 : 128;         // x rows in LDS
    static constexpr int XAHEAD = TAPS9 ? 2 : 0;           // x is written this many stages ahead of gy's stage
    static constexpr int XBEHIND = TAPS9 ? 2 : 0;          // ... and a stage reads x rows this many stages behind it
    static constexpr int NSET = 3;                         // register sets of loads in flight
    static constexpr size_t LDS = (size_t)NBG * 32 * GRS + (size_t)RING * XRS;
    static constexpr int GP = TAPS9 ? 1 : 2, XP = TAPS9 ? 1 : 2;   // passes of 256 lanes x 8 floats per stage
};

template <bool TAPS9>
__device__ __forceinline__ void wgrad_body(const WgradP& p, int bx) {
    using C = WgCfg<TAPS9>;
    constexpr int COT = C::COT, CIT = C::CIT, GRS = C::GRS, XRS = C::XRS, NBG = C::NBG, RING = C::RING, NSET = C::NSET;
    constexpr int XAHEAD = C::XAHEAD, XBEHIND = C::XBEHIND, GP = C::GP, XP = C::XP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const gbuf = smem;
    char* const xring = smem + NBG * 32 * GRS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (position chunk, output-channel tile, input-channel tile)
    int b = bx;
    const int cit = b % p.ci_tiles; b /= p.ci_tiles;
    const int cot = b % p.co_tiles; b /= p.co_tiles;
    const int kc = b;
    const int s0 = kc * p.spc, s1 = min(p.nstage, s0 + p.spc);
    const int co0 = cot * COT, ci0 = cit * CIT;
    const int V = p.N * p.IMG;

    if (wave >= 4) {
        // ============================================================== loaders: global -> registers -> split -> LDS
        const int lt = tid - 256;
        const float upg = __builtin_ldexpf(1.f, split_exponent(amax_rec(p.amax_g, lane)));
        const float upx = __builtin_ldexpf(1.f, split_exponent(amax_rec(p.amax_x, lane)));
        const uintptr_t zpage = reinterpret_cast<uintptr_t>(p.zeros);
        // lane -> (row in pass, 8-channel group): COT / 8 (CIT / 8) lanes per row
        constexpr int GL = COT / 8, XL = CIT / 8, GROWS = 256 / GL, XROWS = 256 / XL;
        const int g_r = lt / GL, g_c = lt % GL, x_r = lt / XL, x_c = lt % XL;
        struct Set { float4 g[GP][2]; float4 x[XP][2]; unsigned xok; };
        Set ld[NSET];
        // BatchNorm + ReLU in front of x: a lane's eight channels never change, their factors are loaded once
        float pmu[8] = {0, 0, 0, 0, 0, 0, 0, 0}, psa[8] = {1, 1, 1, 1, 1, 1, 1, 1}, psb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool pre = p.pre_x != nullptr;
        if (pre && ci0 + 8 * x_c < p.Cin) {
            const float4* f = reinterpret_cast<const float4*>(p.pre_x + ci0 + 8 * x_c);
            const float4 m0 = f[0], m1 = f[1], a0 = f[p.Cin / 4], a1 = f[p.Cin / 4 + 1], b0 = f[p.Cin / 2], b1 = f[p.Cin / 2 + 1];
            pmu[0] = m0.x; pmu[1] = m0.y; pmu[2] = m0.z; pmu[3] = m0.w; pmu[4] = m1.x; pmu[5] = m1.y; pmu[6] = m1.z; pmu[7] = m1.w;
            psa[0] = a0.x; psa[1] = a0.y; psa[2] = a0.z; psa[3] = a0.w; psa[4] = a1.x; psa[5] = a1.y; psa[6] = a1.z; psa[7] = a1.w;
            psb[0] = b0.x; psb[1] = b0.y; psb[2] = b0.z; psb[3] = b0.w; psb[4] = b1.x; psb[5] = b1.y; psb[6] = b1.z; psb[7] = b1.w;
        }
        // A lane's rows move 32 virtual positions per step: (n, y, x) of each is kept and advanced (no division per load).
        struct Pos { int n, y, x; };
        const int RI = p.IMG / p.IP, q32 = 32 / p.IP, r32 = 32 - q32 * p.IP;
        auto pos_of = [&](int v) {               // floor semantics for negative v (rows in front of the first image)
            int n = v / p.IMG, r = v - n * p.IMG;
            if (r < 0) { r += p.IMG; --n; }
            Pos o; o.n = n; o.y = r / p.IP; o.x = r - o.y * p.IP;
            return o;
        };
        auto advance = [&](Pos& o) {
            o.x += r32; o.y += q32;
            if (o.x >= p.IP) { o.x -= p.IP; ++o.y; }
            while (o.y >= RI) { o.y -= RI; ++o.n; }
        };
        auto pixel = [&](const Pos& o) {         // pixel index (n H + y) W + x, or -1 for a gap / outside the batch
            return (o.n >= 0 && o.n < p.N && o.y < p.H && o.x < p.W) ? (o.n * p.H + o.y) * p.W + o.x : -1;
        };
        const int jb = s0 - XBEHIND - XAHEAD;
        Pos pg[GP], px_[XP];
#pragma unroll
        for (int q = 0; q < GP; ++q) pg[q] = pos_of(32 * jb + g_r + GROWS * q);
#pragma unroll
        for (int q = 0; q < XP; ++q) px_[q] = pos_of(32 * (jb + XAHEAD) + x_r + XROWS * q);
        auto issue = [&](Set& L, int sg) {        // loads of the NEXT step (gy stage sg, x stage sg + XAHEAD); advances the rows
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const int row = g_r + GROWS * q;
                const int px = (row < 32 && sg >= s0 && sg < s1) ? pixel(pg[q]) : -1;
                const int ch = co0 + 8 * g_c;
#ifdef NW_WABL_NOLOAD   // ablation builds (tools/bench_wgrad.hip)
                const bool ok = false; (void)px; (void)ch;
#else
                const bool ok = px >= 0 && ch < p.Cout;
#endif
                const float4* src = reinterpret_cast<const float4*>(ok ? reinterpret_cast<uintptr_t>(p.gy + (size_t)px * p.ldg + ch) : zpage);
                L.g[q][0] = src[0];
                L.g[q][1] = src[1];
                advance(pg[q]);
            }
            L.xok = 0;
#pragma unroll
            for (int q = 0; q < XP; ++q) {
                const int row = x_r + XROWS * q;
                const int px = row < 32 ? pixel(px_[q]) : -1;
                const int ch = ci0 + 8 * x_c;
#ifdef NW_WABL_NOLOAD
                const bool ok = false; (void)px; (void)ch;
#else
                const bool ok = px >= 0 && ch < p.Cin;
#endif
                if (p.rr_s > 0) {   // a row run: two 4-channel pixels per lane, each inside or outside the input row on its own
                    const Pos& o = px_[q];
                    // "channel" ch = 32 ky + 4 kx + ci: kernel row ky's run, pixel kx of it (all kernel rows in ONE problem: gy is
                    // read once per 128 of them, not once per kernel row)
                    const int yi = p.rr_s * o.y + p.rr_row0 + (ch >> 5), xa = p.rr_s * o.x + p.rr_col0 + ((ch & 31) >> 2);
                    const bool okr = ok && yi >= 0 && yi < p.rr_H;
                    const float* rowp = p.x + ((size_t)(o.n * p.rr_H + yi) * p.rr_W) * 4;
                    const bool oka = okr && xa >= 0 && xa < p.rr_W, okb = okr && xa + 1 >= 0 && xa + 1 < p.rr_W;
                    L.x[q][0] = *reinterpret_cast<const float4*>(oka ? reinterpret_cast<uintptr_t>(rowp + 4 * xa) : zpage);
                    L.x[q][1] = *reinterpret_cast<const float4*>(okb ? reinterpret_cast<uintptr_t>(rowp + 4 * xa + 4) : zpage);
                    L.xok |= okr ? (1u << q) : 0u;
                    advance(px_[q]);
                    continue;
                }
                const float4* src = reinterpret_cast<const float4*>(ok ? reinterpret_cast<uintptr_t>(p.x + (size_t)px * p.ldx + ch) : zpage);
                L.x[q][0] = src[0];
                L.x[q][1] = src[1];
                L.xok |= ok ? (1u << q) : 0u;
                advance(px_[q]);
            }
        };
        auto split8 = [](const float4& a, const float4& bq, float up, uint4& hv, uint4& lv) {
            unsigned hh[4], ll[4];
            const float xv[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int u = 0; u < 2; ++u)
                asm volatile(
                    "v_fma_mixlo_f16 %0, %4, %8, 0\n"
                    "v_fma_mixlo_f16 %1, %6, %8, 0\n"
                    "v_fma_mixhi_f16 %0, %5, %8, 0\n"
                    "v_fma_mixhi_f16 %1, %7, %8, 0\n"
                    "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n"
                    "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel_hi:[0,0,1]\n"
                    "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
                    "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
                    "s_nop 0"
                    : "=&v"(hh[2 * u]), "=&v"(hh[2 * u + 1]), "=&v"(ll[2 * u]), "=&v"(ll[2 * u + 1])
                    : "v"(xv[4 * u]), "v"(xv[4 * u + 1]), "v"(xv[4 * u + 2]), "v"(xv[4 * u + 3]), "v"(up));
            hv = make_uint4(hh[0], hh[1], hh[2], hh[3]);
            lv = make_uint4(ll[0], ll[1], ll[2], ll[3]);
        };
        // k-major split image: a row holds, per 32 channels, [16 h | 16 h | 16 l | 16 l] as four 32-byte units; the unit
        // index is XOR-ed with unit_swz(row) inside its aligned group of eight (transposed reads then spread over the banks)
        auto store8 = [](char* rowp, int row, int cgrp, const uint4& hv, const uint4& lv) {
            const int u = 4 * (cgrp >> 2) + ((cgrp >> 1) & 1), f = unit_swz(row);
            *reinterpret_cast<uint4*>(rowp + (((u ^ f)) << 5) + ((cgrp & 1) << 4)) = hv;
            *reinterpret_cast<uint4*>(rowp + ((((u + 2) ^ f)) << 5) + ((cgrp & 1) << 4)) = lv;
        };
        auto write = [&](const Set& L, int sg, int sx) {
#ifdef NW_WABL_NOCVT
            if (L.g[0][0].x != 12345.678f) return;
#endif
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const int row = g_r + GROWS * q;
                if (row < 32 && sg >= s0) {
                    uint4 hv, lv;
                    split8(L.g[q][0], L.g[q][1], upg, hv, lv);
                    store8(gbuf + ((unsigned)sg % NBG) * (32 * GRS) + row * GRS, row, g_c, hv, lv);
                }
            }
#pragma unroll
            for (int q = 0; q < XP; ++q) {
                const int row = x_r + XROWS * q;
                if (row < 32) {
                    uint4 hv, lv;
                    if (pre) {   // the forward's relu((x - mean) a + beta), NaN-keeping; a position that is no pixel stays zero
                        const float xs[8] = {L.x[q][0].x, L.x[q][0].y, L.x[q][0].z, L.x[q][0].w, L.x[q][1].x, L.x[q][1].y, L.x[q][1].z, L.x[q][1].w};
                        float xv[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float t = __builtin_fmaf(xs[k] - pmu[k], psa[k], psb[k]);
                            xv[k] = t < 0.f ? 0.f : t;
                        }
                        split8(make_float4(xv[0], xv[1], xv[2], xv[3]), make_float4(xv[4], xv[5], xv[6], xv[7]),
                               ((L.xok >> q) & 1) ? upx : 0.f, hv, lv);
                    } else
                    split8(L.x[q][0], L.x[q][1], upx, hv, lv);
                    const int rr = (32 * sx + row) & (RING - 1);
                    store8(xring + rr * XRS, rr, x_c, hv, lv);
                }
            }
        };
        // Loader step j writes gy stage j and x stage j + XAHEAD.  Consumer stage s multiplies fragments it read during
        // stage s - 1 and reads those of stage s + 1 (gy(s + 1), x stages s + 1 - XBEHIND .. s + 1 + XAHEAD), so it may
        // start once the steps <= s + 1 are written; step s + 2 is written meanwhile (its gy buffer, and the ring rows it
        // overwrites -- x stage s + 2 + XAHEAD - RING / 32 --, are not read then).  The loads of step j + NSET are issued
        // into the registers step j leaves.
#pragma unroll
        for (int u = 0; u < NSET; ++u) issue(ld[u], jb + u);
        for (int j = jb; j < s1; j += NSET) {
#pragma unroll
            for (int u = 0; u < NSET; ++u) {
                const int jj = j + u;
                if (jj >= s1) break;
                write(ld[u], jj, jj + XAHEAD);
                issue(ld[u], jj + NSET);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (jj >= s0 + 1) __builtin_amdgcn_s_barrier();   // releases consumer stage jj - 1
            }
        }
        __builtin_amdgcn_s_barrier();                          // releases the last stage (its successor's rows are never used)
        __builtin_amdgcn_s_barrier();                          // the consumers' barrier behind their last stage
        return;
    }

    // ================================================================== consumers
    const int i = lane & 15, g = lane >> 4;
    const int tq = i >> 2, tp = i & 3;
    const float ig = __builtin_ldexpf(1.f, -split_exponent(wave_max(fmaxf(fmaxf(reinterpret_cast<const float4*>(p.amax_g)[lane].x,
                          reinterpret_cast<const float4*>(p.amax_g)[lane].y), fmaxf(reinterpret_cast<const float4*>(p.amax_g)[lane].z,
                          reinterpret_cast<const float4*>(p.amax_g)[lane].w)))));
    const float ix = __builtin_ldexpf(1.f, -split_exponent(wave_max(fmaxf(fmaxf(reinterpret_cast<const float4*>(p.amax_x)[lane].x,
                          reinterpret_cast<const float4*>(p.amax_x)[lane].y), fmaxf(reinterpret_cast<const float4*>(p.amax_x)[lane].z,
                          reinterpret_cast<const float4*>(p.amax_x)[lane].w)))));
    const float unscale = ig * ix;
    auto tr4 = [](const char* addr) {
        return __builtin_bit_cast(halfx4w, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4w*)addr));
    };
    // transposed read of 32 k x 16 columns: lane 16 g + 4 q + p supplies row 8 g + q (+ 4), bytes 8 p .. 8 p + 7 of the
    // block's 32-byte unit; `unit` = index of the h (or l) unit of the 16-column block inside the row
    auto tr8 = [&](const char* base, int rs, int row0, int ring_mask, int unit) {
        const int ra = (row0 + 8 * g + tq) & ring_mask, rb = (row0 + 8 * g + tq + 4) & ring_mask;
        const halfx4w a = tr4(base + ra * rs + ((unit ^ unit_swz(ra)) << 5) + 8 * tp);
        const halfx4w bq = tr4(base + rb * rs + ((unit ^ unit_swz(rb)) << 5) + 8 * tp);
        return half8{a[0], a[1], a[2], a[3], bq[0], bq[1], bq[2], bq[3]};
    };
#ifdef NW_WABL_NOMFMA
    auto mm = [](const half8& a, const half8& bq, f32x4 c) { c[0] += (float)a[0] * (float)bq[0]; return c; };
#else
    auto mm = [](const half8& a, const half8& bq, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq, c, 0, 0, 0); };
#endif
    auto unit_of = [](int col0, bool low) { return 4 * (col0 >> 5) + ((col0 >> 4) & 1) + (low ? 2 : 0); };

    constexpr int NACC = TAPS9 ? 18 : 16;
    f32x4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 1, wn = wave & 1;               // WIDE: 64 x 64 quadrant
    constexpr int NA = TAPS9 ? 2 : 4, NBF = TAPS9 ? 9 : 4; // gy fragments per set; x fragments per stage (one per tap / column block)
    constexpr int NSLOT = TAPS9 ? 3 : 4;                   // x fragments kept in registers
    half8 ah[2][NA], al[2][NA], bh[NSLOT], bl[NSLOT];
    // Fragments roll: the gy fragments of stage s + 1 are read into the other set at the start of stage s; x fragment f
    // sits in slot f % NSLOT and the slot is refilled with fragment f + NSLOT (of the next stage, past the last one) as
    // soon as its products have been issued -- every LDS read has NSLOT - 1 fragments' worth of matrix work to land under.
    // Addresses of the transposed reads.  A lane's first row inside a 32-row stage is lr = 8 g + q, its second lr + 4; the
    // unit swizzle depends on the row modulo 16 only, so everything but the stage's position in the ring is a per-lane
    // constant: byte = ((32 st + k) & (RING - 1)) * XRS + c with k, c fixed per fragment; the l half sits 64 bytes from
    // the h half (unit + 2 = unit ^ 2: h units have bit 1 clear).  (Computed per read this was ~500 VALU instructions per
    // stage beside 54 MFMAs.)
    const int lr = 8 * g + tq;
    auto swz_c = [&](int row, int unit) { return ((unit ^ unit_swz(row)) << 5) + 8 * tp; };
    int ga[NA], gb_[NA];                                    // gy: offsets inside a stage buffer
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int u = unit_of(TAPS9 ? 16 * a : 64 * wm + 16 * a, false);
        ga[a] = lr * GRS + swz_c(lr, u);
        gb_[a] = (lr + 4) * GRS + swz_c(lr + 4, u);
    }
    int xk[NBF], xa[NBF], xb[NBF];                          // x: row offset k, constants of the two reads
#pragma unroll
    for (int f = 0; f < NBF; ++f) {
        const int shift = TAPS9 ? (f / 3 - 1) * p.IP + (f % 3 - 1) : 0;
        const int u = unit_of(TAPS9 ? 16 * wave : 64 * wn + 16 * f, false);
        xk[f] = shift + lr;
        xa[f] = swz_c(shift + lr, u);
        xb[f] = swz_c(shift + lr + 4, u);
    }
    auto rd8 = [&](const char* pa, const char* pb) {
#ifdef NW_WABL_NOFRAG
        return half8{(_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1};
#endif
        const halfx4w a = tr4(pa), bq = tr4(pb);
        return half8{a[0], a[1], a[2], a[3], bq[0], bq[1], bq[2], bq[3]};
    };
    auto load_a1 = [&](half8& h, half8& l, int a, int st) {
        const char* gb = gbuf + ((unsigned)st % NBG) * (32 * GRS);
        h = rd8(gb + ga[a], gb + gb_[a]);
        l = rd8(gb + (ga[a] ^ 64), gb + (gb_[a] ^ 64));
    };
    auto load_a = [&](half8 (&h)[NA], half8 (&l)[NA], int st) {
#pragma unroll
        for (int a = 0; a < NA; ++a) load_a1(h[a], l[a], a, st);
    };
    auto load_b = [&](int slot, int f, int st) {
        const int r = 32 * st + xk[f];
        const int oa = ((r & (RING - 1)) * XRS) + xa[f], ob = (((r + 4) & (RING - 1)) * XRS) + xb[f];
        bh[slot] = rd8(xring + oa, xring + ob);
        bl[slot] = rd8(xring + (oa ^ 64), xring + (ob ^ 64));
    };
    auto stage = [&](half8 (&h)[NA], half8 (&l)[NA], half8 (&hn)[NA], half8 (&ln)[NA], int st) {
        if (TAPS9) load_a(hn, ln, st + 1);
#pragma unroll
        for (int f = 0; f < NBF; ++f) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int k = TAPS9 ? 2 * f + a : 4 * a + f;
                acc[k] = mm(l[a], bh[f % NSLOT], acc[k]);
                acc[k] = mm(h[a], bl[f % NSLOT], acc[k]);
                acc[k] = mm(h[a], bh[f % NSLOT], acc[k]);
                if (!TAPS9 && f == NBF - 1) {   // WIDE keeps ONE set of gy fragments: each is refilled behind its last product
                    __builtin_amdgcn_sched_barrier(0);
                    load_a1(h[a], l[a], a, st + 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // (keeps the refill behind the products that still need the old fragment)
            if (f + NSLOT < NBF) load_b(f % NSLOT, f + NSLOT, st);
            else load_b(f % NSLOT, f + NSLOT - NBF, st + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        tile_barrier();
    };
    __builtin_amdgcn_s_barrier();                          // stages s0 and s0 + 1 (and the x rows around them) are in LDS
    load_a(ah[0], al[0], s0);
#pragma unroll
    for (int f = 0; f < NSLOT; ++f) load_b(f, f, s0);
    if (TAPS9) {
        for (int s = s0; s < s1; s += 2) {
            stage(ah[0], al[0], ah[1], al[1], s);
            if (s + 1 >= s1) break;
            stage(ah[1], al[1], ah[0], al[0], s + 1);
        }
    } else {
        for (int s = s0; s < s1; ++s) stage(ah[0], al[0], ah[0], al[0], s);
    }
    // ---- partial tile: part[kc][co][t][ci]; acc[.][e] of lane (i, g) = C[row 4 g + e of its co block][column i of its ci block]
    float* out = p.part + (size_t)kc * p.Cout * p.T * p.Cin;
    if (TAPS9) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + 16 * a + 4 * g + e, ci = ci0 + 16 * wave + i;
                    if (co < p.Cout && ci < p.Cin) out[((size_t)co * p.T + t) * p.Cin + ci] = acc[2 * t + a][e] * unscale;
                }
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + 64 * wm + 16 * a + 4 * g + e, ci = ci0 + 64 * wn + 16 * c + i;
                    if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[4 * a + c][e] * unscale;
                }
    }
}

template <bool TAPS9>
__global__ __launch_bounds__(512, 1) void nw_conv_wgrad_kernel(const WgradP p) { wgrad_body<TAPS9>(p, blockIdx.x); }

// Several independent weight gradients in ONE launch (the kernel arguments hold up to 32 problems and the first workgroup of
// each): the 14x14 and 7x7 layers of a dense block give 12-38 workgroups each -- launched one after the other they leave
// most of the chip idle for ~20 us apiece; their gradients are not on the backward's critical path, so a block's layers are
// collected and run together.
constexpr int WG_BATCH = 32;
struct WgradBatch {
    int n;
    int first[WG_BATCH + 1];
    WgradP p[WG_BATCH];
};
template <bool TAPS9>
__global__ __launch_bounds__(512, 1) void nw_conv_wgrad_batch_kernel(const WgradBatch bt) {
    int prob = 0;
    for (int k = 1; k < bt.n; ++k)
        if ((int)blockIdx.x >= bt.first[k]) prob = k;      // (wave-uniform: scalar compares)
    wgrad_body<TAPS9>(bt.p[prob], (int)blockIdx.x - bt.first[prob]);
}

struct RedJob { const float* part; float* dw; long long total4; int ks; int first; int T, Cin; };   // T > 1: dw is written as (Cout, Cin, T)
struct RedBatch { int n; RedJob j[WG_BATCH]; };

// dw[idx] = sum_k part[k][idx] in chunk order: 64 float4 columns x 16 chunk lanes per workgroup -- lane j adds the chunks
// j, j + 16, ... (coalesced across the columns), the 16 partial sums are added in lane order through LDS (deterministic)
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ dw, int64_t total4, int ks,
                                                  int bx, int T = 1, int Cin = 0) {
    __shared__ float4 sh[16][64];
    const int col = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int64_t idx = (int64_t)bx * 64 + col;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < total4)
        for (int k = j; k < ks; k += 16) {
            const float4 v = reinterpret_cast<const float4*>(part)[(int64_t)k * total4 + idx];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    sh[j][col] = a;
    __syncthreads();
    if (j == 0 && idx < total4) {
        for (int l = 1; l < 16; ++l) {
            const float4 v = sh[l][col];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (T <= 1) {
            reinterpret_cast<float4*>(dw)[idx] = a;
        } else {   // partials are (Cout, T, Cin); the caller wants torch's (Cout, Cin, KH, KW): four strided stores (small tensors)
            const int64_t e = idx * 4, row = e / Cin, ci = e - row * Cin, co = row / T, t = row - co * T;
            float* o = dw + (co * Cin + ci) * T + t;
            o[0] = a.x; o[T] = a.y; o[2 * T] = a.z; o[3 * T] = a.w;
        }
    }
}
__global__ __launch_bounds__(1024) void nw_conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                     int64_t total4, int ks) {
    wgrad_reduce_body(part, dw, total4, ks, blockIdx.x);
}
__global__ __launch_bounds__(1024) void nw_conv_wgrad_reduce_batch_kernel(const RedBatch rb) {
    int job = 0;
    for (int k = 1; k < rb.n; ++k)
        if ((int)blockIdx.x >= rb.j[k].first) job = k;
    wgrad_reduce_body(rb.j[job].part, rb.j[job].dw, rb.j[job].total4, rb.j[job].ks, (int)blockIdx.x - rb.j[job].first, rb.j[job].T,
                      rb.j[job].Cin);
}

struct WgPlan {
    bool taps9;
    int co_tiles, ci_tiles, nstage, ks, spc;
    int IP, IMG;
};

bool wgrad_plan(int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad,
                WgPlan* pl, int target_wgs = 256) {
    if (n <= 0 || H <= 0 || W <= 0 || stride != 1 || KH != KW || (KH != 1 && KH != 3) || pad != (KH - 1) / 2) return false;
    if (Cin % 8 || Cout % 8 || Cin < 8 || Cout < 8) return false;
    pl->taps9 = KH == 3;
    if (pl->taps9 && W > 62) return false;                 // a tap shift (W + 2 rows) must stay within two 32-row stages
    pl->IP = (int)(pl->taps9 ? W + 1 : W);
    pl->IMG = (int)(pl->taps9 ? (H + 1) * (W + 1) : H * W);
    const int64_t V = n * pl->IMG;
    if (V >= (1LL << 30) || n * H * W * Cin >= (1LL << 31) || n * H * W * Cout >= (1LL << 31)) return false;
    pl->nstage = (int)((V + 31) / 32);
    const int cot = pl->taps9 ? 32 : 128, cit = pl->taps9 ? 64 : 128;
    pl->co_tiles = (int)((Cout + cot - 1) / cot);
    pl->ci_tiles = (int)((Cin + cit - 1) / cit);
    const int64_t tiles = (int64_t)pl->co_tiles * pl->ci_tiles;
    // ~one workgroup per CU, and no more chunks than the output tile is worth: the partial tiles cross memory twice
    // (written, read by the reduction), 2 ks Cout T Cin floats against (Cout + Cin) 32 nstage read by the product
    int64_t ks = (target_wgs + tiles - 1) / tiles;   // (a batch of problems shares the chip: fewer chunks each)
    // at least 4 (3x3: 8, a chunk starts with four steps of rows around its first stage) stages per chunk: small planes are
    // latency-bound and want many workgroups
    int64_t minst = pl->taps9 ? 8 : 4;   // (16 / 8 until round 3: K4 19.05 -> 18.83 ms with twice the workgroups on the small planes)
    if (knob(KNOB_WGRAD_MIN_STAGES) > 0) minst = pl->taps9 ? knob(KNOB_WGRAD_MIN_STAGES) : (knob(KNOB_WGRAD_MIN_STAGES) + 1) / 2;
    const int64_t maxks = (pl->nstage + minst - 1) / minst;
    if (ks > maxks) ks = maxks;
    if (ks < 1) ks = 1;
    pl->spc = (int)((pl->nstage + ks - 1) / ks);
    pl->ks = (int)((pl->nstage + pl->spc - 1) / pl->spc);
    return true;
}

}  // namespace
}  // namespace nw

extern "C" int nw_conv2d_nhwc_wgrad_supported(int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                                              int64_t stride, int64_t pad) {
    nw::WgPlan pl;
    return nw::wgrad_plan(n, H, W, Cin, Cout, KH, KW, stride, pad, &pl) ? 1 : 0;
}

extern "C" size_t nw_conv2d_nhwc_wgrad_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH,
                                                       int64_t KW, int64_t stride, int64_t pad) {
    nw::WgPlan pl;
    if (!nw::wgrad_plan(n, H, W, Cin, Cout, KH, KW, stride, pad, &pl)) return 0;
    return (size_t)pl.ks * Cout * KH * KW * Cin * sizeof(float);
}

extern "C" const void* nw_conv_zero_page(void);

extern "C" int nw_conv2d_nhwc_wgrad_f16x2(const float* x, const float* amax_x, const float* gy, const float* amax_g, float* dw,
                                          void* workspace, size_t workspace_bytes, int64_t n, int64_t H, int64_t W, int64_t Cin,
                                          int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, int64_t ldx,
                                          int64_t ldg, void* stream) {
    using namespace nw;
    WgPlan pl;
    if (!wgrad_plan(n, H, W, Cin, Cout, KH, KW, stride, pad, &pl)) return NW_ERR_UNSUPPORTED;
    if (ldx == 0) ldx = Cin;
    if (ldg == 0) ldg = Cout;
    if (ldx < Cin || ldg < Cout || ldx % 4 || ldg % 4) return NW_ERR_INVALID_ARG;
    if (n * H * W * ldx >= (1LL << 31) || n * H * W * ldg >= (1LL << 31)) return NW_ERR_UNSUPPORTED;   // 32-bit element offsets with the row strides
    if (!x || !amax_x || !gy || !amax_g || !dw) return NW_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(dw) |
         reinterpret_cast<uintptr_t>(amax_x) | reinterpret_cast<uintptr_t>(amax_g) | reinterpret_cast<uintptr_t>(workspace)) & 15)
        return NW_ERR_INVALID_ARG;
    const size_t need = nw_conv2d_nhwc_wgrad_workspace_bytes(n, H, W, Cin, Cout, KH, KW, stride, pad);
    if (!workspace || workspace_bytes < need) return NW_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    WgradP p;
    p.x = x; p.amax_x = amax_x; p.gy = gy; p.amax_g = amax_g; p.pre_x = nullptr;
    p.rr_s = p.rr_H = p.rr_W = p.rr_row0 = p.rr_col0 = 0;
    p.part = pl.ks == 1 ? dw : static_cast<float*>(workspace);
    p.zeros = static_cast<const float4*>(nw_conv_zero_page());
    if (!p.zeros) return NW_ERR_LAUNCH;
    p.N = (int)n; p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout; p.T = (int)(KH * KW); p.KW = (int)KW; p.pad = (int)pad;
    p.ldx = (int)ldx; p.ldg = (int)ldg;
    p.IP = pl.IP; p.IMG = pl.IMG; p.nstage = pl.nstage; p.ks = pl.ks; p.spc = pl.spc; p.co_tiles = pl.co_tiles; p.ci_tiles = pl.ci_tiles;
    const unsigned grid = (unsigned)((int64_t)pl.ks * pl.co_tiles * pl.ci_tiles);
    if (pl.taps9) {
        static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(nw_conv_wgrad_kernel<true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgCfg<true>::LDS) == hipSuccess;
        if (!attr) return NW_ERR_LAUNCH;
        hipLaunchKernelGGL(nw_conv_wgrad_kernel<true>, dim3(grid), dim3(512), WgCfg<true>::LDS, st, p);
    } else {
        static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(nw_conv_wgrad_kernel<false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgCfg<false>::LDS) == hipSuccess;
        if (!attr) return NW_ERR_LAUNCH;
        hipLaunchKernelGGL(nw_conv_wgrad_kernel<false>, dim3(grid), dim3(512), WgCfg<false>::LDS, st, p);
    }
    if (pl.ks > 1) {
        const int64_t total4 = Cout * KH * KW * Cin / 4;
        hipLaunchKernelGGL(nw_conv_wgrad_reduce_kernel, dim3((unsigned)((total4 + 63) / 64)), dim3(1024), 0, st,
                           static_cast<const float*>(workspace), dw, total4, pl.ks);
    }
    NW_CHECK_LAUNCH();
    return NW_OK;
}

// ---- a batch of weight gradients (nw_wgrad_job, include/nwhead_hip.h)
static int batch_target_wgs(int64_t njobs) {
    const int k = nw::knob(nw::KNOB_WGRAD_BATCH_WGS);
    if (k > 0) return k > 256 ? 256 : k;   // (the jobs' workspace slices are sized for at most 256 workgroups per job: wgrad_job_ws)
    // the jobs share the chip: ~1024 workgroups in all, 64..256 per job (K4: 17.36-17.42 ms at 64 per job, 17.66-17.70 at 256:
    // fewer chunks = smaller partial tiles and a shorter reduce)
    const int64_t t = 1024 / (njobs > 0 ? njobs : 1);
    return (int)(t < 64 ? 64 : (t > 256 ? 256 : t));
}
static size_t wgrad_job_ws(const nw_wgrad_job& j) {   // (sized for the largest split: one chunk per ~256 / tiles workgroups)
    size_t b = nw_conv2d_nhwc_wgrad_workspace_bytes(j.n, j.H, j.W, j.Cin, j.Cout, j.KH, j.KW, j.stride, j.pad);
    const size_t one = (size_t)(j.Cout * j.KH * j.KW * j.Cin) * sizeof(float);
    if (b < one) b = one;
    return (b + 255) & ~(size_t)255;
}

extern "C" size_t nw_conv2d_nhwc_wgrad_batch_workspace_bytes(const nw_wgrad_job* jobs, int64_t njobs) {
    size_t total = 0;
    for (int64_t k = 0; jobs && k < njobs; ++k) total += wgrad_job_ws(jobs[k]);
    return total;
}

extern "C" int nw_conv2d_nhwc_wgrad_batch_f16x2(const nw_wgrad_job* jobs, int64_t njobs, void* workspace, size_t workspace_bytes,
                                                void* stream) {
    using namespace nw;
    if (njobs < 0 || (njobs > 0 && !jobs)) return NW_ERR_INVALID_ARG;
    if (njobs == 0) return NW_OK;
    if (workspace_bytes < nw_conv2d_nhwc_wgrad_batch_workspace_bytes(jobs, njobs) || (workspace_bytes && !workspace) ||
        (reinterpret_cast<uintptr_t>(workspace) & 15))
        return NW_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float4* zeros = static_cast<const float4*>(nw_conv_zero_page());
    if (!zeros) return NW_ERR_LAUNCH;
    static const bool attr9 = hipFuncSetAttribute(reinterpret_cast<const void*>(nw_conv_wgrad_batch_kernel<true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgCfg<true>::LDS) == hipSuccess;
    static const bool attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(nw_conv_wgrad_batch_kernel<false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgCfg<false>::LDS) == hipSuccess;
    if (!attr9 || !attr1) return NW_ERR_LAUNCH;
    const int tgt = batch_target_wgs(njobs);
    // validate everything first: nothing is launched for a batch with a bad job
    for (int64_t k = 0; k < njobs; ++k) {
        const nw_wgrad_job& j = jobs[k];
        WgPlan pl;
        if (!wgrad_plan(j.n, j.H, j.W, j.Cin, j.Cout, j.KH, j.KW, j.stride, j.pad, &pl, tgt)) return NW_ERR_UNSUPPORTED;
        const int64_t ldx = j.ldx ? j.ldx : j.Cin, ldg = j.ldg ? j.ldg : j.Cout;
        if (!j.x || !j.amax_x || !j.gy || !j.amax_g || !j.dw || ldx < j.Cin || ldg < j.Cout || ldx % 4 || ldg % 4) return NW_ERR_INVALID_ARG;
        if (j.n * j.H * j.W * ldg >= (1LL << 31) || (!j.rowrun_stride && j.n * j.H * j.W * ldx >= (1LL << 31))) return NW_ERR_UNSUPPORTED;
        if ((reinterpret_cast<uintptr_t>(j.x) | reinterpret_cast<uintptr_t>(j.gy) | reinterpret_cast<uintptr_t>(j.dw) |
             reinterpret_cast<uintptr_t>(j.amax_x) | reinterpret_cast<uintptr_t>(j.amax_g) | reinterpret_cast<uintptr_t>(j.pre_x)) & 15)
            return NW_ERR_INVALID_ARG;
        if (j.rowrun_stride) {   // row runs: the plan of a 1x1 problem over gy's grid with 32 "channels" (8 pixels x 4)
            if (j.rowrun_stride < 0 || j.KH != 1 || j.Cin % 32 || j.pre_x || j.in_H <= 0 || j.in_W <= 0 ||
                j.n * j.in_H * j.in_W * 4 >= (1LL << 31))
                return NW_ERR_INVALID_ARG;
        }
    }
    char* wsp = static_cast<char*>(workspace);
    for (int pass = 0; pass < 2; ++pass) {                   // 3x3 problems, then 1x1 problems: two kernels
        const bool want9 = pass == 0;
        WgradBatch bt;
        RedBatch rb;
        bt.n = 0; bt.first[0] = 0; rb.n = 0;
        int red_wgs = 0;
        auto flush = [&]() {
            if (bt.n) {
                if (want9) hipLaunchKernelGGL(nw_conv_wgrad_batch_kernel<true>, dim3((unsigned)bt.first[bt.n]), dim3(512), WgCfg<true>::LDS, st, bt);
                else hipLaunchKernelGGL(nw_conv_wgrad_batch_kernel<false>, dim3((unsigned)bt.first[bt.n]), dim3(512), WgCfg<false>::LDS, st, bt);
            }
            if (rb.n) hipLaunchKernelGGL(nw_conv_wgrad_reduce_batch_kernel, dim3((unsigned)red_wgs), dim3(1024), 0, st, rb);
            bt.n = 0; bt.first[0] = 0; rb.n = 0; red_wgs = 0;
        };
        size_t off = 0;
        for (int64_t k = 0; k < njobs; ++k) {
            const nw_wgrad_job& j = jobs[k];
            const size_t wsz = wgrad_job_ws(j);
            char* jws = wsp + off;
            off += wsz;
            WgPlan pl;
            wgrad_plan(j.n, j.H, j.W, j.Cin, j.Cout, j.KH, j.KW, j.stride, j.pad, &pl, tgt);
            if (pl.taps9 != want9) continue;
            WgradP& q = bt.p[bt.n];
            q.x = j.x; q.amax_x = j.amax_x; q.gy = j.gy; q.amax_g = j.amax_g; q.pre_x = j.pre_x;
            q.rr_s = (int)j.rowrun_stride; q.rr_H = (int)j.in_H; q.rr_W = (int)j.in_W; q.rr_row0 = (int)j.row0; q.rr_col0 = (int)j.col0;
            const bool oihw = j.out_oihw != 0 && j.KH * j.KW > 1;   // torch's weight layout: written by the reduce kernel
            q.part = (pl.ks == 1 && !oihw) ? j.dw : reinterpret_cast<float*>(jws);
            q.zeros = zeros;
            q.N = (int)j.n; q.H = (int)j.H; q.W = (int)j.W; q.Cin = (int)j.Cin; q.Cout = (int)j.Cout; q.T = (int)(j.KH * j.KW);
            q.KW = (int)j.KW; q.pad = (int)j.pad;
            q.ldx = (int)(j.ldx ? j.ldx : j.Cin); q.ldg = (int)(j.ldg ? j.ldg : j.Cout);
            q.IP = pl.IP; q.IMG = pl.IMG; q.nstage = pl.nstage; q.ks = pl.ks; q.spc = pl.spc; q.co_tiles = pl.co_tiles; q.ci_tiles = pl.ci_tiles;
            bt.first[bt.n + 1] = bt.first[bt.n] + pl.ks * pl.co_tiles * pl.ci_tiles;
            ++bt.n;
            if (pl.ks > 1 || oihw) {
                RedJob& r = rb.j[rb.n];
                r.part = reinterpret_cast<const float*>(jws); r.dw = j.dw; r.total4 = j.Cout * j.KH * j.KW * j.Cin / 4; r.ks = pl.ks;
                r.T = oihw ? (int)(j.KH * j.KW) : 1; r.Cin = (int)j.Cin;
                r.first = red_wgs;
                red_wgs += (int)((r.total4 + 63) / 64);
                ++rb.n;
            }
            if (bt.n == WG_BATCH) flush();
        }
        flush();
    }
    NW_CHECK_LAUNCH();
    return NW_OK;
}
