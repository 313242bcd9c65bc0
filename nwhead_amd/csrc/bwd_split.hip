// bwd_split.hip -- the two products of the head's backward on the fp16 matrix cores (gfx950 / MI355X only).
//
//     gq[b,k] = sum_j A[b,j] s[j,k] + 2 rq[b] q[b,k]          gs[j,k] = sum_b A[b,j] q[b,k] + 2 rs[j] s[j,k]
// (closed form of autograd through nwhead/nw.py:276-289 and nwhead/kernel.py:13-44, see backward.hip.)
//
// Same arithmetic as the forward (tile_f16.h): every fp32 operand is a pair of fp16 numbers x = h + l and a
// product of two of them is three v_mfma_f32_16x16x32_f16 (hl, lh, hh; exact products, fp32 accumulation).
// The scaling has to respect the contraction index: a power of two per ROW of an operand can only be undone
// after the sum if that row index is an OUTPUT index.  With e_j the split exponent of support row j
// (s'_j = s_j 2^e_j, the forward's bank format) and E_b that of row b of A^ = A 2^-e_j:
//     gq[b,k] = 2^-E_b     sum_j A'[b,j] s'[j,k]              A'[b,j] = A[b,j] 2^(E_b - e_j)
//     gs[j,k] = 2^(e_j-G)  sum_b A'[b,j] q''[b,k]             q''[b,k] = q[b,k] 2^(G - E_b),  G global
// so ONE split image of A' serves both products: row-major reads for the first (k = j runs along its rows) and
// transposed reads (ds_read_b64_tr_b16) for the second (k = b runs across them).  s' and q'' are read transposed
// in both.  The coefficient kernel (backward.hip) writes A' and 2^-E_b; nw_bwd_qsplit_kernel writes q'' and 2^-G.
//
// One kernel, C[m,n] = sum_k X(m,k) Y[k,n]:  128 x 64 output tile per 512-thread workgroup: four multiplying waves of 64 x 32
// (eight accumulator blocks, 24 MFMAs per 32 k) and four loader waves; operands HBM/L2 -> LDS by global_load_lds_dwordx4 into a
// three-stage ring (24 KB per stage, two workgroups per CU), one barrier per stage.  LDS images:
//   X row-major ("MK"): [128 m][128 B = 32 h | 32 l], 16-byte slots XOR-swizzled by (m >> 1) & 7 (the forward's image);
//   X or Y k-major ("KM"/"KN"): [32 k][W bytes], W = 512 (128 m) / 256 (64 n); the 32-byte units (16 columns of h or
//     of l) XOR-swizzled by f(k) = (k & 3) | ((k >> 3) & 1) << 2: the 8 rows one 32-lane half of a transposed read
//     touches land on 8 different groups of 8 banks.
// K is split over blockIdx.z when M x N alone cannot fill the chip; partial tiles are summed in chunk order.
#include "nw_internal.h"
#include "tile_dma.h"
#include <cstdlib>

namespace nw {
namespace {

typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

constexpr int XM = 128, XN = 64, XK = 32;
constexpr int X_BYTES = XM * 128, Y_BYTES = XK * XN * 4, XST_BYTES = X_BYTES + Y_BYTES;   // 16 KB + 8 KB

// XNBUF = 3 / 2: two / three workgroups per CU (many short K loops, the second product), fragments single-buffered --
// the other workgroups' waves cover the LDS latency.  XNBUF = 6: one workgroup per CU (few long K loops, the first product),
// fragments double-buffered in registers: the reads of stage s+1 are issued under the MFMAs of stage s.
#ifdef XG_DIAG   // diagnostic build only (tools/bench_xgemm.hip): s_memtime stamps per workgroup
__device__ unsigned long long nw_diag_x[8 * 4096];
#define XG_STAMP(k) do { if (lane == 0 && (wave == 0 || wave == 4)) nw_diag_x[8 * (blockIdx.x & 4095) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XG_STAMP(k) do { } while (0)
#endif
constexpr int XOUT_LD = 68;   // floats per row of the staged output tile (272 B: the four row groups of a store hit two bank sets)

template <bool X_KM, bool FUSE, int XNBUF>
__global__ __launch_bounds__(512, XNBUF == 2 ? 6 : (XNBUF == 3 ? 4 : 1)) void nw_xgemm_kernel(
    const char* __restrict__ X, int64_t x_row_bytes, int x_rows, const char* __restrict__ Y, int64_t y_row_bytes,
    int y_rows, float* __restrict__ out, const float* __restrict__ fac, int fac_inverse,
    const float* __restrict__ gfac, const float* __restrict__ rowscale, const float* __restrict__ Xo, int M, int Nn,
    int K, int k_chunk, int gx, int gy, int gz, int rows_per_group, XgemmReduce pend) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool PIPE = XNBUF >= 6;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The first pend.blocks workgroups finish an EARLIER product of the stream (its K-split partial tiles, summed in
    // chunk order): the reduction rides along instead of being a launch of its own behind this one.
    if ((int)blockIdx.x < pend.blocks) {
        const int64_t total4 = pend.M * pend.Nn / 4;
        const int64_t idx = (int64_t)blockIdx.x * 512 + tid;
        if (idx < total4) {
            const int64_t m = (idx * 4) / pend.Nn;
            float4 a = *reinterpret_cast<const float4*>(pend.part + idx * 4);
            for (int c = 1; c < pend.nchunks; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(pend.part + ((int64_t)c * pend.M * pend.Nn) + idx * 4);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            float f = pend.fac[m];
            f = (pend.fac_inverse ? 1.f / f : f) * (pend.gfac ? *pend.gfac : 1.f);
            const float rsc = 2.f * pend.rowscale[m];
            const float4 x = *reinterpret_cast<const float4*>(pend.Xo + idx * 4);
            a.x = __builtin_fmaf(rsc, x.x, a.x * f); a.y = __builtin_fmaf(rsc, x.y, a.y * f);
            a.z = __builtin_fmaf(rsc, x.z, a.z * f); a.w = __builtin_fmaf(rsc, x.w, a.w * f);
            *reinterpret_cast<float4*>(pend.out + idx * 4) = a;
        }
        return;
    }
    const int bid = blockIdx.x - pend.blocks;   // (pend.blocks % 8 == 0: the XCD of a tile does not move)
    // Workgroup -> tile, XCD-aware: consecutive workgroup ids go round the 8 XCDs (each with its own L2), so the
    // tiles that share an operand are given ids 8 apart.  A "row" r = z gy + m is one X tile (all gx n-tiles read
    // it); rows_per_group rows (all m-tiles of one K chunk when K is split: they share the Y tiles) form a group,
    // and group number 8 t + x runs on XCD x.
    const int xcd = bid & 7, slot = bid >> 3;
    const int per_group = gx * rows_per_group;
    const int gi = (slot / per_group) * 8 + xcd, t_in = slot % per_group;
    const int r = gi * rows_per_group + t_in / gx;
    if (r >= gy * gz) return;
    const int bx = t_in % gx, by = r % gy, bz = r / gy;
    const int m0 = by * XM, n0 = bx * XN;
    const int kb = bz * k_chunk, ke = min(K, kb + k_chunk);
    const int nst = (ke - kb + XK - 1) / XK;
    // waves 0-3 multiply (64 x 32 of the tile each), waves 4-7 move data: a wave that has LDS-DMAs in flight gets an
    // s_waitcnt vmcnt(0) from hipcc in front of every LDS read (the DMA is a pending LDS write it cannot tell from the
    // buffer being read), which serialises load and compute.  The loaders' registers are otherwise idle: they fetch
    // the tile of the rank-one term (2 rowscale[m] Xo[m,n]) before the loop and finish the output after it.
    const int lw = wave & 3;
    float* stage_out = reinterpret_cast<float*>(smem);   // [128][XOUT_LD] floats over the ring, once the loop is done
    if (wave == 0) XG_STAMP(0);

    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int ltid = tid - 256;
        float4 xo[8];
        float ff[8], rs2[8], gf = 1.f;
        // DMA sources (per lane; a stage advances them by 32 k).  X: 16 instructions per stage, loader wave lw issues
        // t = 4 lw .. 4 lw + 3; Y: 8 instructions (4 k-rows of 256 bytes each), t = 2 lw, 2 lw + 1
        const char* xsrc[4];
        int xrow[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = 4 * lw + u;
            if (X_KM) {
                const int row = 2 * t + (lane >> 5), sl = lane & 31;
                const int f = (row & 3) | (((row >> 3) & 1) << 2);
                xrow[u] = row;
                xsrc[u] = X + (int64_t)m0 * 4 + ((((sl >> 1) ^ f) << 1) | (sl & 1)) * 16;
            } else {
                const int row = 8 * t + (lane >> 3), sl = lane & 7;
                xrow[u] = 0;
                xsrc[u] = X + (int64_t)min(m0 + row, x_rows - 1) * x_row_bytes + (sl ^ ((row >> 1) & 7)) * 16;
            }
        }
        const char* ysrc[2];
        int yrow[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = 2 * lw + u;
            const int row = 4 * t + (lane >> 4), sl = lane & 15;
            const int f = (row & 3) | (((row >> 3) & 1) << 2);
            yrow[u] = row;
            ysrc[u] = Y + (int64_t)n0 * 4 + ((((sl >> 1) ^ f) << 1) | (sl & 1)) * 16;
        }
        auto issue = [&](int s) {
            char* st = smem + (unsigned)(s % XNBUF) * XST_BYTES;
            const int k0 = kb + s * XK;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const char* src = X_KM ? xsrc[u] + (int64_t)min(k0 + xrow[u], x_rows - 1) * x_row_bytes
                                       : xsrc[u] + (int64_t)k0 * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(st + 1024 * (4 * lw + u)), 16, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const char* src = ysrc[u] + (int64_t)min(k0 + yrow[u], y_rows - 1) * y_row_bytes;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(st + X_BYTES + 1024 * (2 * lw + u)), 16, 0, 0);
            }
        };
        constexpr int PER = 6;
        // stages that may still be in flight at a barrier: all but the one (PIPE: two) the consumers read next
        constexpr int FLY = (XNBUF - (PIPE ? 3 : 2)) * PER;
#pragma unroll
        for (int s = 0; s < XNBUF - 1; ++s)
            if (s < nst) issue(s);
        // the rank-one term's operands, behind the first DMAs: loads return in order, so the counted waits below
        // cover them too (the first barrier waits for them: they arrive with the first stages, not in front of them)
        if (FUSE) {   // loads only: nothing here may wait for a result before the DMAs are on their way
            if (gfac) gf = *gfac;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int m = min(m0 + 16 * t + (ltid >> 4), M - 1), n = n0 + 4 * (ltid & 15);
                xo[t] = *reinterpret_cast<const float4*>(Xo + (int64_t)m * Nn + (n < Nn ? n : 0));
                ff[t] = fac[m];
                rs2[t] = rowscale[m];
            }
        }
        if (nst >= XNBUF - 1) wait_vmcnt<FLY>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nst; ++s) {
#ifndef XG_NO_DMA   // (timing ablations of tools/bench_xgemm.hip: wrong results)
            if (s + XNBUF - 1 < nst) issue(s + XNBUF - 1);   // into the buffer the consumers left at the last barrier
#endif
            if (s + XNBUF - 1 < nst) wait_vmcnt<FLY>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        XG_STAMP(5);
        __builtin_amdgcn_s_barrier();   // the consumers have staged the accumulators
        // finish: 16 lanes per output row (256 bytes), 16 rows per pass
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = 16 * t + (ltid >> 4), c4 = ltid & 15;
            const int m = m0 + row, n = n0 + 4 * c4;
            float4 v = *reinterpret_cast<const float4*>(stage_out + row * XOUT_LD + 4 * c4);
            if (FUSE) {
                const float f = (fac_inverse ? 1.f / ff[t] : ff[t]) * gf, r2 = 2.f * rs2[t];   // f: powers of two, exact
                v.x = __builtin_fmaf(r2, xo[t].x, v.x * f); v.y = __builtin_fmaf(r2, xo[t].y, v.y * f);
                v.z = __builtin_fmaf(r2, xo[t].z, v.z * f); v.w = __builtin_fmaf(r2, xo[t].w, v.w * f);
            }
            if (m < M && n < Nn)
                *reinterpret_cast<float4*>(out + ((int64_t)(FUSE ? 0 : bz) * M + m) * Nn + n) = v;
        }
        XG_STAMP(4);
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    const int wm = (lw >> 1) * 64, wn = (lw & 1) * 32;
    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed reads: lane 16 g + 4 q + p supplies row 8 g + q (+ 4 for the second read), bytes 8 p .. 8 p + 7 of the
    // block's 32-byte unit
    const int tq = i >> 2, tp = i & 3;
    const int tf32 = (tq | ((g & 1) << 2)) * 32;
    const int yoff = (8 * g + tq) * (XN * 4) + 8 * tp;          // + ((unit * 32) ^ tf32), + 4 rows = 4 * 256 bytes
    const int xoff_t = (8 * g + tq) * (XM * 4) + 8 * tp;
    const int rsw = (i >> 1) & 7;
    const int xoff_r = (wm + i) * 128;

    auto tr8 = [&](const char* base, int off, int row4_bytes) {
        const halfx4 a = __builtin_bit_cast(halfx4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) fp16x4*)(base + off)));
        const halfx4 b = __builtin_bit_cast(halfx4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) fp16x4*)(base + off + row4_bytes)));
        return half8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };
    auto mm = [](const half8& a, const half8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); };
    struct Frag {
        half8 yh[2], yl[2], xh[4], xl[4];
    };
    auto load_frags = [&](Frag& f, int s) {
        const char* Xs = smem + (unsigned)(s % XNBUF) * XST_BYTES;
        const char* Ys = Xs + X_BYTES;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int e0 = wn + 16 * b;                                   // first column of the block within the tile
            const int uh = 4 * (e0 >> 5) + ((e0 >> 4) & 1);               // its 32-byte unit of h; l is two units on
            f.yh[b] = tr8(Ys, yoff + ((uh * 32) ^ tf32), 4 * XN * 4);
            f.yl[b] = tr8(Ys, yoff + (((uh + 2) * 32) ^ tf32), 4 * XN * 4);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (X_KM) {
                const int e0 = wm + 16 * a;
                const int uh = 4 * (e0 >> 5) + ((e0 >> 4) & 1);
                f.xh[a] = tr8(Xs, xoff_t + ((uh * 32) ^ tf32), 4 * XM * 4);
                f.xl[a] = tr8(Xs, xoff_t + (((uh + 2) * 32) ^ tf32), 4 * XM * 4);
            } else {
                const float4* row = reinterpret_cast<const float4*>(Xs + xoff_r + a * 16 * 128);
                f.xh[a] = __builtin_bit_cast(half8, row[g ^ rsw]);
                f.xl[a] = __builtin_bit_cast(half8, row[(4 + g) ^ rsw]);
            }
        }
    };
    auto mfma_stage = [&](const Frag& f) {
        // small terms first, the dominant h*h product last
#ifndef XG_ONE_MFMA
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = mm(f.xl[a], f.yh[b], acc[a][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = mm(f.xh[a], f.yl[b], acc[a][b]);
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = mm(f.xh[a], f.yh[b], acc[a][b]);
    };
    auto stage_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    __builtin_amdgcn_s_barrier();   // stage 0 (PIPE: and 1) has landed
    XG_STAMP(1);
    if (PIPE) {
        // the 16 reads of the next stage go one-to-one between the first MFMAs of the current one (left to itself the
        // scheduler issues them after most of the MFMAs and the wait at the barrier exposes their latency)
        auto interleave = [&]() {
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      // the remaining MFMAs
        };
        Frag f0, f1;
        load_frags(f0, 0);
        int s = 0;
        for (; s + 2 < nst; s += 2) {
            load_frags(f1, s + 1);
            mfma_stage(f0);
            interleave();
            stage_end();
            load_frags(f0, s + 2);
            mfma_stage(f1);
            interleave();
            stage_end();
        }
        for (; s < nst; ++s) {   // one or two stages left, nothing further to prefetch
            if (s + 1 < nst) load_frags(f1, s + 1);
            mfma_stage(f0);
            stage_end();
            f0 = f1;
        }
    } else {
        Frag f;
        for (int s = 0; s < nst; ++s) {
            load_frags(f, s);
            mfma_stage(f);
            stage_end();
        }
    }
    XG_STAMP(2);
    // stage the accumulators for the loaders: acc[a][b][e] of lane (i, g) = C[wm + 16 a + 4 g + e][wn + 16 b + i]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                stage_out[(wm + 16 * a + 4 * g + e) * XOUT_LD + wn + 16 * b + i] = acc[a][b][e];
    stage_end();
    XG_STAMP(3);
}

// C[m,n] = f(m) sum_chunks part[c][m,n] + 2 rowscale[m] Xo[m,n]   (n in float4s; Nn % 4 == 0)
__global__ __launch_bounds__(256) void nw_xgemm_reduce_kernel(const float* __restrict__ part, int nchunks,
                                                               const float* __restrict__ fac, int fac_inverse,
                                                               const float* __restrict__ gfac,
                                                               const float* __restrict__ rowscale,
                                                               const float* __restrict__ Xo, float* __restrict__ out,
                                                               int64_t M, int64_t Nn) {
    const int64_t total4 = M * Nn / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int64_t m = (idx * 4) / Nn;
    float4 a = *reinterpret_cast<const float4*>(part + idx * 4);
    for (int c = 1; c < nchunks; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)c * M * Nn) + idx * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float f = fac[m];
    f = (fac_inverse ? 1.f / f : f) * (gfac ? *gfac : 1.f);
    const float rsc = 2.f * rowscale[m];
    const float4 x = *reinterpret_cast<const float4*>(Xo + idx * 4);
    a.x = __builtin_fmaf(rsc, x.x, a.x * f); a.y = __builtin_fmaf(rsc, x.y, a.y * f);
    a.z = __builtin_fmaf(rsc, x.z, a.z * f); a.w = __builtin_fmaf(rsc, x.w, a.w * f);
    *reinterpret_cast<float4*>(out + idx * 4) = a;
}

int xgemm_target_wgs() {
    const int v = knob(KNOB_XGEMM_WGS);
    return v > 0 ? v : 256;
}

}  // namespace

XgemmPlan xgemm_plan(int64_t M, int64_t Nn, int64_t K) {
    const int64_t tiles = ((M + XM - 1) / XM) * ((Nn + XN - 1) / XN);
    const int64_t target = xgemm_target_wgs();
    int64_t want = tiles >= target ? 1 : (target + tiles - 1) / tiles;
    const int64_t maxc = K / 128 > 1 ? K / 128 : 1;   // at least four stages per chunk
    if (want > maxc) want = maxc;
    int64_t kc = (K + want - 1) / want;
    kc = (kc + XK - 1) / XK * XK;
    XgemmPlan p;
    p.k_chunk = (int)kc;
    p.nchunks = (int)((K + kc - 1) / kc);
    return p;
}

// x_km = false: X is (M rows, k along the row) -- true: X is (k rows, m along the row).  Both operands are split-row
// images (tile_f16.h); K-direction padding must be zero in ONE of them and finite in the other; rows are clamped to
// x_rows / y_rows, columns past a row's end are read from the bytes that follow (the caller pads the buffers by
// XGEMM_TAIL_BYTES) and only feed outputs that are never stored.
int launch_xgemm(bool x_km, const float* X, int64_t ldx, int64_t x_rows, const float* Y, int64_t ldy, int64_t y_rows,
                 float* part, const float* fac, int fac_inverse, const float* gfac, const float* rowscale,
                 const float* Xo, float* out, int64_t M, int64_t Nn, int64_t K, hipStream_t st, XgemmReduce* defer,
                 const XgemmReduce* pending) {
    const XgemmPlan p = xgemm_plan(M, Nn, K);
    XgemmReduce pend = {};
    if (pending && pending->blocks > 0) pend = *pending;
    if (defer) *defer = XgemmReduce{};
    const int64_t gx = (Nn + XN - 1) / XN, gy = (M + XM - 1) / XM, gz = p.nchunks;
    const int64_t rpg = gz >= 8 ? gy : 1, groups = (gy * gz + rpg - 1) / rpg;
    const int64_t nwg = (groups + 7) / 8 * 8 * gx * rpg;
    if (nwg + pend.blocks > 0x7fffffffLL || gy * gz > 0x7fffffffLL) return NW_ERR_INVALID_ARG;
    const dim3 grid((unsigned)(nwg + pend.blocks));
    // ring depth: deep and one workgroup per CU when there are at most ~one workgroup per CU anyway and K is long
    const int nbuf_env = knob(KNOB_XGEMM_NBUF) == KNOB_UNSET ? 0 : knob(KNOB_XGEMM_NBUF);
    const bool deep = nbuf_env ? nbuf_env == 6 : (gx * gy * gz <= 320 && p.k_chunk >= 6 * XK);
    // more than two workgroups per CU's worth of tiles: two stages each, so that three are resident per CU and the whole
    // grid runs in one round (second product at T, 632 workgroups of 8 stages: 17.7 us against 20.3 with three stages)
    const bool thin = nbuf_env ? nbuf_env == 2 : (!deep && gx * gy * gz > 512);
    const size_t lds = (size_t)(deep ? 6 : thin ? 2 : 3) * XST_BYTES;
    const char* Xc = reinterpret_cast<const char*>(X);
    const char* Yc = reinterpret_cast<const char*>(Y);
#define NW_XG1(KM, FUSE, NB, OUT)                                                                                         \
    hipLaunchKernelGGL((nw_xgemm_kernel<KM, FUSE, NB>), grid, dim3(512), lds, st, Xc, ldx * 4, (int)x_rows, Yc, ldy * 4,   \
                       (int)y_rows, OUT, fac, fac_inverse, gfac, rowscale, Xo, (int)M, (int)Nn, (int)K, p.k_chunk,       \
                       (int)gx, (int)gy, (int)gz, (int)rpg, pend)
#define NW_XG(KM, FUSE, OUT)                                                                                              \
    do {                                                                                                                  \
        if (deep) NW_XG1(KM, FUSE, 6, OUT);                                                                               \
        else if (thin) NW_XG1(KM, FUSE, 2, OUT);                                                                          \
        else NW_XG1(KM, FUSE, 3, OUT);                                                                                    \
    } while (0)
    if (p.nchunks == 1) {
        if (x_km) NW_XG(true, true, out); else NW_XG(false, true, out);
    } else {
        if (x_km) NW_XG(true, false, part); else NW_XG(false, false, part);
        const int64_t total4 = M * Nn / 4;
        const int64_t rblocks = ((total4 + 511) / 512 + 7) / 8 * 8;
        if (defer && rblocks <= 4096) {   // a later launch on this stream finishes this product
            defer->part = part; defer->fac = fac; defer->gfac = gfac; defer->rowscale = rowscale; defer->Xo = Xo;
            defer->out = out; defer->M = M; defer->Nn = Nn; defer->nchunks = p.nchunks; defer->fac_inverse = fac_inverse;
            defer->blocks = (int)rblocks;
        } else {
            hipLaunchKernelGGL(nw_xgemm_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, part,
                               p.nchunks, fac, fac_inverse, gfac, rowscale, Xo, out, M, Nn);
        }
    }
#undef NW_XG
#undef NW_XG1
    NW_CHECK_LAUNCH();
    return NW_OK;
}

}  // namespace nw
