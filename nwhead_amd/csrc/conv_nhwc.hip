// conv_nhwc.hip -- 2-D convolution of the backbones as an implicit GEMM on the fp16 matrix cores at fp32-grade
// accuracy (gfx950 / MI355X only).  Replaces F.conv2d at the call sites of model/resnet.py:31-66, :147-156, :178-190
// and model/densenet.py:33-60, :82-91 (forward; the data gradient of a stride-1 convolution is the same kernel on the
// flipped, transposed weight).
//
//     y[n, yo, xo, co] = post( bias[co] + sum_{ky,kx,ci} W[co, ky, kx, ci] x[n, s yo + ky - p, s xo + kx - p, ci] [+ res] )
//
// Layout: activations fp32 NHWC (torch channels_last), Cin % 32 == 0, Cout % 32 == 0.  Arithmetic: the head's
// (tile_f16.h): every fp32 operand is a pair of fp16 numbers x = h + l, a product of two operands is three
// v_mfma_f32_16x16x32_f16 (hl, lh, hh), accumulated in fp32.
//   * weights: split ONCE per weight update (nw_split_rows_f16x2 over the (Cout, KH KW Cin) matrix a channels_last
//     weight is: one power of two per output channel -- an OUTPUT index, undone in the store);
//   * activations: split IN FLIGHT by the loader waves (global_load_dwordx4 -> 2^e x -> h, l -> ds_write_b128) with ONE
//     power of two per tensor, from a bound on max|x| the producer of x left in `amax_in` (any upper bound is legal;
//     the convolution's own store does an atomicMax into `amax_out`, so a chain of convolutions never reads a tensor
//     twice).  Per-pixel scales would not do: the taps of one output read pixels with different scales.
//
// One persistent 512-thread workgroup per CU: waves 0-3 multiply (tile = 128 or 256 output pixels x 32 / 64 / 128
// output channels; the output channels are the A operand so that a lane ends up with four consecutive channels of one
// pixel: 16-byte stores), waves 4-7 move data and run ahead across tile boundaries.
//   PATCH mode (3x3, stride 1, padding 1): per 32 input channels the pixels around the tile go to LDS ONCE -- a range
//     of a virtual padded raster (rows W + 1 apart, one zero row between images: the zeros are written by the loader,
//     the nine taps are nine constant row shifts and no border test exists in the loop) -- and serve nine stages; the
//     weights of each (channel chunk, tap) stage come by LDS-DMA.
//   GATHER mode (everything else: 1x1, strided, other kernel sizes): every stage gathers the pixels of its tap.
#include "nw_internal.h"
#include "tile_dma.h"
#include <cstdlib>

namespace nw {
namespace {

// what the loaders read in place of a pixel that does not exist (padding, tile tail): selecting the ADDRESS costs two
// v_cndmask per load, selecting the data one per element
__device__ float4 nw_conv_zeros[2];

struct ConvP {
    const float* x;
    const float* amax_in;    // CV_AMAX_SLOTS floats, their maximum bounds max|x|
    const char* ws;          // split weight rows, (Cout, T * Cin) floats-worth of bytes
    const float* wscale;     // (Cout,) 2^-e of the weight rows
    const float* bias;       // nullable
    const float* res;        // nullable (M, Cout)
    float* y;
    float* amax_out;         // nullable: CV_AMAX_SLOTS floats
    float* moments;          // nullable: per (tile row, wave row) group and output channel the count, mean, M2, minimum and
                             //   maximum of y (part[(k G + group) Cout + co], k = 0 .. 4, G = mtiles WN): BatchNorm's statistics
                             //   of y, and the bound on |relu(bn(y))| its consumer's split needs, without a pass over it
    int pre_nrec;            // PRE: 0: amax_in bounds |x'| itself; n > 0: amax_in holds n amax records of the RAW tensor x (an
                             //   inference caller: the producers' records) and the kernel derives the bound from the table,
                             //   max_c |a_c| (A + |mean_c|) + max(beta_c, 0) with A = the records' maximum
    const float* pre;        // nullable: [3][Cin] = mean | a | beta -- the convolution reads x' = relu((x - mean) a + beta), a
                             //   BatchNorm + ReLU applied by the loaders on the way into LDS (round 4: model/densenet.py:36-45's
                             //   norm1-relu1-conv1 / norm2-relu2-conv2 without the tensors between them); amax_in bounds |x'|
    const float4* zeros;     // 32 bytes of zeros (what a loader reads for a pixel that does not exist)
    // nullable (bnb_part): y is the gradient of relu(batch_norm(bnb_x)) -- the epilogue also leaves, per pixel group and
    // channel, sum g and sum g xhat with g = y [bn(x) > 0]: BatchNorm's backward statistics without a pass over (x, y)
    const float* bnb_x; const float* bnb_mean; const float* bnb_invstd; const float* bnb_gamma; const float* bnb_beta;
    float* bnb_part;
    int bnb_ldx;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, relu;
    int ldx, ldy;            // floats between consecutive pixels of x / y (>= Cin / Cout: a channel window of a wider tensor)
    int IP, IMG;             // virtual raster of PATCH mode: row stride W + pad, image stride (H + pad) IP
    int M, mtiles, ntiles;
    int macc;                // moments: ONE group per (workgroup, wave row), merged over the workgroup's tiles (ntiles == 1)
};

constexpr int CV_GATHER = 0, CV_PATCH = 1, CV_ROWRUN = 2, CV_ROWRUN4 = 3;
constexpr int CV_AMAX_SLOTS = 256;   // floats of an `amax` record: one partial maximum per workgroup of its producer

template <int NA, int NB, int WM, int MODE, bool PRE = false>
struct ConvCfg {
    static constexpr bool PATCH = MODE == CV_PATCH, ROWRUN = MODE == CV_ROWRUN || MODE == CV_ROWRUN4;
    static_assert(!(PRE && ROWRUN), "the few-channel stems have no BatchNorm in front");
    static constexpr int WN = 4 / WM;
    static constexpr int BN = 16 * NA * WM, BM = 16 * NB * WN;
    static constexpr int NIW = BN / 32;                           // weight DMAs per loader wave per stage
    static constexpr int EMAX = PATCH ? BM + 192 : BM;            // activation entries (pixels) per buffer
    static constexpr int NPASS = EMAX / 64;
    static constexpr int NACT = (MODE == CV_ROWRUN ? 8 : 2) * NPASS;   // global loads per lane per chunk
    static constexpr int TI = PATCH ? 9 : 1;                      // stages per activation chunk
    // register sets of activation loads in flight: a one-stage chunk (GATHER) needs several to cover the memory latency
    static constexpr int NSET = PATCH ? 1 : (MODE == CV_ROWRUN ? 2 : 4);
    static constexpr int UNR = PATCH ? 9 : NSET;                  // the loaders' loop is unrolled over one period of their issue order
    // weight ring depth (PRE: the BatchNorm table of the input channels -- 12 bytes per channel -- sits in LDS behind the rings)
    static constexpr int NWR = MODE == CV_ROWRUN ? 4 : (BN == 128 ? (PATCH ? 4 : (PRE ? 5 : 6)) : (PATCH ? 8 : 6));
    static constexpr int AH = NWR - 1;                            // weight stages issued ahead
    static constexpr int WST = BN * 128, PB = EMAX * 128;
    // activation buffers: a chunk is written two stages before its first stage; the first stage of a TILE is read at
    // its own start (no fragment prefetch across tiles), so with one-stage chunks the buffer two chunks back may still
    // be in use: three buffers in GATHER mode
    static constexpr int NPB = PATCH ? 2 : 3;
    static constexpr size_t LDS = (size_t)NWR * WST + NPB * (size_t)PB;
    static_assert(EMAX % 64 == 0 && LDS <= 160 * 1024, "tile shape");
    static_assert((AH - 2) * ((PATCH ? 0 : NACT) + NIW) + (PATCH ? NACT : 0) < 64 && 9 * NIW < 64 &&
                  NSET * NIW + (NSET - 1) * NACT < 64, "vmcnt is a 6-bit field");
};

// max of an amax record (CV_AMAX_SLOTS floats), by one wave
__device__ __forceinline__ float amax_read(const float* rec, int lane) {
    const float4 v = reinterpret_cast<const float4*>(rec)[lane];
    return wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
}
// PRE with raw records (ConvP::pre_nrec > 0): the bound on |relu((x - mean) a + beta)| from the table in LDS and the maximum A of
// the n records; every wave computes the same value (no hand-over)
__device__ __forceinline__ float pre_bound_from_table(const float* tab_lds, int Cin, const float* rec, int nrec, int lane) {
    float A = 0.f;
    for (int k = lane; k < 64 * nrec; k += 64) {
        const float4 v = reinterpret_cast<const float4*>(rec)[k];
        A = fmaxf(A, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    A = wave_max(A);
    float b = 0.f;
    for (int c = lane; c < Cin; c += 64)
        b = fmaxf(b, __builtin_fmaf(fabsf(tab_lds[Cin + c]), A + fabsf(tab_lds[c]), fmaxf(tab_lds[2 * Cin + c], 0.f)));
    b = wave_max(b);
    return (A < INFINITY && b < INFINITY) ? b : INFINITY;
}
// 16-byte slot of piece s (0-3: the h halves of channels 8 s .. 8 s + 7, 4-7: the l halves) inside the 128-byte LDS row of
// activation entry e.  ds_read_b128 serves a wave in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... --
// i.e. all 16 entries e0 + i once, with piece g for eight of them and g ^ 1 for the other eight; bit 0 of the slot follows the
// piece (so the two halves of a group never meet) and the upper bits rotate with e / 2 (four same-parity entries of either
// half: four distinct values) -- conflict-free for EVERY e0.  The XOR swizzle s ^ (e / 2 & 7) of the head's tiles is
// conflict-free only for e0 % 4 == 0; a 3x3 tap shifts e0 by dy IP + dx: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE went from
// 2.79 M / 7.35 M to 1.26 M / 5.82 M cycles on the 128 -> 32 3x3 layer at 56 x 56 (the rest: the loaders' stores) -- at the
// same 67 us: the LDS is not what bounds that kernel ("what bounds the narrow tiles").
__device__ __forceinline__ int pslot(int e, int s) { return (s & 1) | ((((s >> 1) ^ (e >> 1)) & 3) << 1); }
// workgroup b of g writes its maximum to slot b and zeros to the slots b + g, b + 2 g, ... it stands in for
__device__ __forceinline__ void amax_write(float* rec, float m, int b, int g) {
    for (int k = b; k < CV_AMAX_SLOTS; k += g) rec[k] = k == b ? m : 0.f;
}

#ifdef NW_CONV_DIAG   // diagnostic build only (tools/bench_conv.hip): s_memtime totals of workgroup phases, [wg][16]
__device__ unsigned long long nw_conv_diag[16 * 1024];
#define NW_CSTAMP(k)                                                                          \
    do {                                                                                      \
        unsigned long long now_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        cd_[k] += now_ - cl_;                                                                 \
        cl_ = now_;                                                                           \
    } while (0)
#else
#define NW_CSTAMP(k)
#endif

// STATS = false: the instantiation for launches that leave neither moments nor BatchNorm backward sums (inference, data
// gradients): none of that code and none of its running registers (their mere presence cost every convolution 1-5 %)
// POST = false: likewise without bias, identity and ReLU (the training path's convolutions have none of them)
// PRE = true: the loaders apply relu((x - mean) a + beta) per input channel (ConvP::pre) in front of the split
template <int NA, int NB, int WM, int MODE, bool STATS, bool POST, bool PRE = false>
__global__ __launch_bounds__(512, 1) void nw_conv_nhwc_kernel(const ConvP p) {
    // (running moments cost 33 registers across the main loop: the 128 x 128 tile has none to spare and keeps them in LDS, 128
    //  bytes per wave and lane row behind the rings, touched by the row's first lane only)
    constexpr bool MACC_LDS = NA == 4 && NB == 4;
    const int p_macc = p.macc;
    float* const p_moments = STATS ? p.moments : nullptr;
    float* const p_bnb_part = STATS ? p.bnb_part : nullptr;
    const float* const p_bias = POST ? p.bias : nullptr;
    const float* const p_res = POST ? p.res : nullptr;
    const int p_relu = POST ? p.relu : 0;
    using C = ConvCfg<NA, NB, WM, MODE, PRE>;
    constexpr bool PATCH = C::PATCH, ROWRUN = C::ROWRUN;
    constexpr int BN = C::BN, BM = C::BM, NIW = C::NIW, NPASS = C::NPASS, NACT = C::NACT, TI = C::TI, NSET = C::NSET;
    constexpr int NWR = C::NWR, AH = C::AH, WST = C::WST, PB = C::PB, NPB = C::NPB, UNR = C::UNR, WN = C::WN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wring = smem;
    char* const pbuf = smem + NWR * WST;
    // PRE: [3][Cin] mean | a | beta of the input channels, copied once by the loader waves (first version: six global loads per
    // lane and chunk beside the two of the activations -- 4x the vector-memory instructions of a 1x1 layer's loaders, +10 us
    // on the 14 x 14 layers)
    char* const ptab = smem + C::LDS + (NA == 4 && NB == 4 ? 4096 : 0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ROWRUN (few input channels, e.g. the 7x7 / 2 stem over RGB): the KW * Cin <= 32 values one kernel ROW reads are
    // contiguous in NHWC memory; they are one 32-wide k chunk (zero-padded, in the weight too) and a stage is a kernel row
    const int T = ROWRUN ? p.KH : p.KH * p.KW, nc = ROWRUN ? 1 : p.Cin >> 5;
    const int ST = nc * T;                                         // stages per tile
    const int CH = ST / TI;                                        // activation chunks per tile
    // tiles of this workgroup: every XCD (workgroups with equal id mod 8) walks a contiguous range, its workgroups
    // take consecutive tiles (neighbouring tiles share halo pixels and weights in the XCD's L2)
    const int total = p.mtiles * p.ntiles;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int tpx = (total + 7) >> 3;
    const int tbeg = xcd * tpx + slot, tend = min(total, (xcd + 1) * tpx);
    const int ntile = tbeg < tend ? (tend - tbeg + nslot - 1) / nslot : 0;
    if (ntile == 0) {
        if (p.amax_out && tid == 0) amax_write(p.amax_out, 0.f, blockIdx.x, gridDim.x);
        if ((p_moments || p_bnb_part) && p_macc) {                 // its groups exist and are empty
            const int G = gridDim.x * WN, nstat = p_moments ? 5 : 2;
            float* dst = p_moments ? p_moments : p_bnb_part;
            for (int k = tid; k < nstat * WN * p.Cout; k += 512) {
                const int w3 = k / p.Cout, c = k - w3 * p.Cout;    // (statistic, wave row)
                const int stat = w3 / WN;                          // (an empty group: count 0, minimum +inf, maximum -inf)
                dst[((size_t)stat * G + blockIdx.x * WN + w3 % WN) * p.Cout + c] = stat == 3 ? INFINITY : (stat == 4 ? -INFINITY : 0.f);
            }
        }
        return;
    }
    const int S = ntile * ST;                                      // stages of this workgroup
    const int QT = ntile * CH;                                     // activation chunks of this workgroup
    const int HW = p.Ho * p.Wo;
    const size_t wrow = ROWRUN ? (size_t)T * 128 : (size_t)T * p.Cin * 4;   // bytes per weight row

    if (wave >= 4) {
        // ============================================================== loaders
        const int lw = wave - 4, lt = tid - 256, lj = lt & 3, le = lt >> 2;
        typedef __attribute__((address_space(3))) char lds_char_;
        if constexpr (PRE) {
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            const int n4 = 3 * p.Cin / 4;
            for (int k = lt; k < n4; k += 256) {
                const u32x4_ v = reinterpret_cast<const u32x4_*>(p.pre)[k];
                const unsigned a_ = (unsigned)(uintptr_t)(lds_char_*)(ptab + 16 * k);
                asm volatile("ds_write_b128 %0, %1" ::"v"(a_), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // (the consumers pass the same barrier at their start)
        }
        const unsigned ptab_lane = PRE ? (unsigned)(uintptr_t)(lds_char_*)(ptab + 32 * lj) : 0u;   // this lane's eight channels of chunk 0
        // ---- weight DMA: stage cursor
        unsigned woff[NIW];
#pragma unroll
        for (int m = 0; m < NIW; ++m) {
            const int R = 8 * (lw + 4 * m) + (lane >> 3);
            woff[m] = (unsigned)((size_t)R * wrow) + (unsigned)(((lane & 7) ^ ((R >> 1) & 7)) << 4);
        }
        int d_tile = 0, d_k = 0, d_s = 0;                          // next stage to issue: tile index, stage in tile, global
        const char* w_tile = p.ws;
        size_t w_off = 0;
        int w_t = 0;
        // Every iteration issues the same instructions (past the end of the run: the weights' first rows into a ring
        // slot nobody reads any more, the page of zeros for the activations), so that the order of issue is periodic
        // and both hipcc's own waits for the register loads and the counted waits below are exact.
        int w_skip = 0, a_skip = 0;                                // dummy issues in front of the first real ones (pipeline fill)
        auto issue_w = [&]() {
            const char* base = p.ws;
            if (w_skip > 0) {                                      // fill: same instructions, stage 0's slot (its own DMA lands later)
                --w_skip;
                char* dst0 = wring;
#ifndef NW_CABL_NODMA
#pragma unroll
                for (int m = 0; m < NIW; ++m)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + woff[m]),
                                                     (__attribute__((address_space(3))) void*)(dst0 + 1024 * (lw + 4 * m)), 16, 0, 0);
#endif
                return;
            }
            if (d_s < S) {
                if (d_k == 0) {                                    // a new tile: its weight rows (one division per tile, not per stage)
                    const int t_id = tbeg + d_tile * nslot;
                    w_tile = p.ws + (size_t)((t_id % p.ntiles) * BN) * wrow;
                    w_off = 0;
                    w_t = 0;
                }
                base = w_tile + w_off;
            }
            char* dst = wring + (d_s % NWR) * WST;
#ifndef NW_CABL_NODMA
#pragma unroll
            for (int m = 0; m < NIW; ++m)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + woff[m]),
                                                 (__attribute__((address_space(3))) void*)(dst + 1024 * (lw + 4 * m)), 16, 0, 0);
#else
            (void)dst;
#endif
            ++d_s;
            // the stage's byte offset inside a weight row, kept incrementally: (t Cin + 32 c) 4 with (c, t) = (k / 9, k % 9) in
            // PATCH order, and simply 128 k otherwise (t nc + c = k)
            if (PATCH) {
                w_off += (size_t)p.Cin * 4;
                if (++w_t == 9) { w_t = 0; w_off -= (size_t)9 * p.Cin * 4 - 128; }
            } else {
                w_off += 128;
            }
            if (++d_k == ST) { d_k = 0; ++d_tile; }
        };
        // ---- activations: chunk cursor
        int a_tile = 0, a_q = 0, a_g = 0;                          // next chunk to LOAD: tile index, chunk in tile, global
        unsigned aoff[NPASS];                                      // PATCH: source offset (floats) of the entry at chunk 0
        unsigned avalid = 0;                                       // bit q: the entry of pass q is a real pixel
        int gy[NPASS], gx[NPASS], gn[NPASS];                       // GATHER: s yo - pad, s xo - pad, n H  (gn < 0: no pixel)
        unsigned gbase[NPASS];                                     // GATHER: element offset of tap (0, 0), chunk 0 of the pass's pixel
        int g_t = 0, g_c = 0, g_dy = 0, g_dx = 0;                  // GATHER: the cursor's (tap, channel chunk), kept incrementally
        auto setup_tile = [&]() {
            const int t_id = tbeg + a_tile * nslot;
            const int m0 = (t_id / p.ntiles) * BM;
            if (PATCH) {
                const int n0 = m0 / HW, r0 = m0 - n0 * HW, y0 = r0 / p.Wo, x0 = r0 - y0 * p.Wo;
                const int lo = n0 * p.IMG + y0 * p.IP + x0;
                avalid = 0;
#pragma unroll
                for (int q = 0; q < NPASS; ++q) {
                    const int idx = lo + le + 64 * q;
                    const int n = idx / p.IMG, r = idx - n * p.IMG, row = r / p.IP, col = r - row * p.IP;
                    const int yi = row - p.pad, xi = col - p.pad;
                    const bool ok = n < p.N && yi >= 0 && xi >= 0 && xi < p.W;
                    aoff[q] = ok ? (unsigned)(((n * p.H + yi) * p.W + xi) * p.ldx + 8 * lj) : 0u;
                    avalid |= ok ? (1u << q) : 0u;
                }
            } else {
#pragma unroll
                for (int q = 0; q < NPASS; ++q) {
                    const int m = m0 + le + 64 * q;
                    const int n = m / HW, r = m - n * HW, yo = r / p.Wo, xo = r - yo * p.Wo;
                    gy[q] = p.stride * yo - p.pad;
                    gx[q] = p.stride * xo - p.pad;
                    gn[q] = m < p.M ? n * p.H : -1;
                    gbase[q] = (unsigned)(((gn[q] + gy[q]) * p.W + gx[q]) * p.ldx + 8 * lj);   // (wraps for a pixel that does not exist: unused then)
                }
            }
        };
        int rdx[8];                                                // ROWRUN: k -> pixel of the run (k / Cin), far negative past the run
        if (MODE == CV_ROWRUN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * lj + j;
                rdx[j] = k < p.KW * p.Cin ? k / p.Cin : -(1 << 20);
            }
        }
        // a pixel's address, or the page of zeros: selected as INTEGERS (two v_cndmask; given the pointers, hipcc turns the
        // select into a branch around two loads and then waits for everything in flight at the join)
        const uintptr_t zpage = reinterpret_cast<uintptr_t>(p.zeros);
        auto pick = [&](bool ok, const void* ptr) {
#ifdef NW_CABL_NOLOAD
            ok = false;
#endif
            return reinterpret_cast<const float4*>(ok ? reinterpret_cast<uintptr_t>(ptr) : zpage);
        };
        // (the activation loads are plain loads: hipcc's own waits order them against the conversions that read their
        //  registers; with the LDS stores of write_a out of its sight those waits come out counted, not vmcnt(0))
        auto ldg2 = [](float4& a, float4& b, const float4* src) { a = src[0]; b = src[1]; };
        auto ldg1 = [](float4& a, const float4* src) { a = src[0]; };
        float4 ld[NSET][NPASS][2];
        unsigned ldvalid[NSET];
        auto issue_a = [&](float4 (&L)[NPASS][2], unsigned& valid) {   // loads of the cursor's chunk
            const bool live = a_skip == 0 && a_g < QT;
            if (a_skip > 0) --a_skip;
            if (live && a_q == 0) setup_tile();
            valid = 0xffffffffu;
            if (PRE) valid = 0;   // pixels that do not exist must come out as ZERO behind the BatchNorm: bit q = pass q holds a real pixel
            if (PATCH) {
                const float* base = p.x + 32 * a_q;
#pragma unroll
                for (int q = 0; q < NPASS; ++q) {
                    const bool ok = live && ((avalid >> q) & 1);
                    if (PRE) valid |= ok ? (1u << q) : 0u;
                    ldg2(L[q][0], L[q][1], pick(ok, base + aoff[q]));
                }
            } else if (MODE == CV_ROWRUN4) {                       // Cin == 4: a pixel is one aligned float4, two pixels per lane
#pragma unroll
                for (int q = 0; q < NPASS; ++q) {
                    const int yi = gy[q] + a_q;
                    const bool rowok = live && gn[q] >= 0 && yi >= 0 && yi < p.H;
                    const int x0 = gx[q] + 2 * lj, x1 = x0 + 1;
                    const bool ok0 = rowok && 2 * lj < p.KW && (unsigned)x0 < (unsigned)p.W;
                    const bool ok1 = rowok && 2 * lj + 1 < p.KW && (unsigned)x1 < (unsigned)p.W;
                    const float4* row = reinterpret_cast<const float4*>(p.x) + (gn[q] + yi) * p.W;
                    ldg1(L[q][0], pick(ok0, row + x0));
                    ldg1(L[q][1], pick(ok1, row + x1));
                }
            } else if (MODE == CV_ROWRUN) {
                valid = 0;                                         // bit 8 q + j: element j of pass q is a real value
#pragma unroll
                for (int q = 0; q < NPASS; ++q) {
                    const int yi = gy[q] + a_q;
                    const bool rowok = live && gn[q] >= 0 && yi >= 0 && yi < p.H;
                    const int base = ((gn[q] + yi) * p.W + gx[q]) * p.Cin + 8 * lj;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool ok = rowok && (unsigned)(gx[q] + rdx[j]) < (unsigned)p.W;
                        v[j] = p.x[ok ? base + j : 0];
                        valid |= ok ? (1u << (8 * q + j)) : 0u;
                    }
                    L[q][0] = make_float4(v[0], v[1], v[2], v[3]);
                    L[q][1] = make_float4(v[4], v[5], v[6], v[7]);
                }
            } else {
                // (tap, chunk) of the cursor without divisions, the tap's offset once per chunk on the scalar unit: the loaders'
                // issue path is what bounds the 1x1 layers (bench_conv stamps, 42 x 14 x 14 x 512 -> 128: 850 of a stage's 2 500
                // ticks went into issuing two loads)
                if (a_q == 0) g_t = g_c = g_dy = g_dx = 0;
                const int dy = g_dy, dx = g_dx;
                const unsigned tapoff = (unsigned)((dy * p.W + dx) * p.ldx + 32 * g_c);
                if (live) {
                    if (++g_c == nc) { g_c = 0; ++g_t; if (++g_dx == p.KW) { g_dx = 0; ++g_dy; } }
                }
#pragma unroll
                for (int q = 0; q < NPASS; ++q) {
                    const int yi = gy[q] + dy, xi = gx[q] + dx;
                    const bool ok = live && gn[q] >= 0 && yi >= 0 && yi < p.H && xi >= 0 && xi < p.W;
                    const unsigned off = gbase[q] + tapoff;
                    if (PRE) valid |= ok ? (1u << q) : 0u;
                    ldg2(L[q][0], L[q][1], pick(ok, p.x + off));
                }
            }
            if (live) {
                ++a_g;
                if (++a_q == CH) { a_q = 0; ++a_tile; }
            }
        };
        // the tensor's scale: one power of two from the bound its producer left (read behind the first loads)
        float up = 1.f;
        auto write_pass = [&](const float4& L0, const float4& L1, unsigned valid, int g, int q, const f32x4 (&F)[PRE ? 6 : 1]) {   // pass q of a loaded chunk -> split -> buffer g % NPB
            char* pb = pbuf + (g % NPB) * PB;
#ifdef NW_CABL_NOCVT
            if (L0.x != 12345.678f) return;
#endif
            {
                const float xs[8] = {L0.x, L0.y, L0.z, L0.w, L1.x, L1.y, L1.z, L1.w};
                float xv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[k] = (MODE == CV_ROWRUN && !((valid >> (8 * q + k)) & 1)) ? 0.f : xs[k];
                float upq = up;
                if constexpr (PRE) {   // BatchNorm + ReLU in front of the split: bn_nhwc.hip's centred form, its NaN-keeping ReLU
                    const float mu[8] = {F[0][0], F[0][1], F[0][2], F[0][3], F[1][0], F[1][1], F[1][2], F[1][3]};
                    const float sa[8] = {F[2][0], F[2][1], F[2][2], F[2][3], F[3][0], F[3][1], F[3][2], F[3][3]};
                    const float sb[8] = {F[4][0], F[4][1], F[4][2], F[4][3], F[5][0], F[5][1], F[5][2], F[5][3]};
                    // (packed fp32 subtract / FMA, v_max_f32: 16 instructions for the 8 elements.  v_max drops a NaN; a NaN anywhere
                    //  in x makes its channel's mean NaN, the table's bound non-finite, and `up` -- the scale of EVERY element --
                    //  NaN: the whole output is NaN, as the unfused passes would leave the loss)
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        const f32x2_ t = __builtin_elementwise_fma(f32x2_{xv[k], xv[k + 1]} - f32x2_{mu[k], mu[k + 1]}, f32x2_{sa[k], sa[k + 1]},
                                                                   f32x2_{sb[k], sb[k + 1]});
                        xv[k] = fmaxf(t.x, 0.f);
                        xv[k + 1] = fmaxf(t.y, 0.f);
                    }
                    upq = ((valid >> q) & 1) ? up : 0.f;           // a pixel that does not exist: h = l = 0
                }
                // h = fp16(2^e x), l = fp16(2^e x - h): two mixed-precision FMAs per element, written by hand (hipcc
                // builds three quarters of them from packed fp32 FMAs and conversions: 22 instructions for these 16).
                // A half-register write is followed by a read of that register no sooner than two instructions later
                // (gfx950's destination-select forwarding hazard; hipcc pads nothing inside an asm statement).
                unsigned hh[4], ll[4];
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
                        : "v"(xv[4 * u]), "v"(xv[4 * u + 1]), "v"(xv[4 * u + 2]), "v"(xv[4 * u + 3]), "v"(upq));
                const int e = le + 64 * q;
                // the LDS stores by hand too: a wave with LDS-DMAs in flight gets an s_waitcnt vmcnt(0) from hipcc in front
                // of every LDS access it can see (it cannot tell the DMA's target from this buffer)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 hv = {hh[0], hh[1], hh[2], hh[3]}, lv = {ll[0], ll[1], ll[2], ll[3]};
                typedef __attribute__((address_space(3))) char lds_char;
                const unsigned ah_ = (unsigned)(uintptr_t)(lds_char*)(pb + e * 128 + (pslot(e, lj) << 4));
                const unsigned al_ = (unsigned)(uintptr_t)(lds_char*)(pb + e * 128 + (pslot(e, 4 + lj) << 4));
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %2, %3" ::"v"(ah_), "v"(hv), "v"(al_), "v"(lv) : "memory");
            }
        };
        // PRE: the factors of the chunk that is written NEXT are read from the LDS table one iteration ahead (hand-written reads:
        // hipcc would wait vmcnt(0) in front of them); the iteration's own s_waitcnt lgkmcnt(0) in front of its barrier covers them
        int w_q = 0;                                                   // chunk (of its tile) whose factors are read next
        f32x4 Fn[PRE ? 6 : 1];
        auto read_factors = [&]() {
            if constexpr (PRE) {
                const int cch = PATCH ? w_q : w_q % nc;
                const unsigned a0 = ptab_lane + 128u * (unsigned)cch, rs = 4u * (unsigned)p.Cin;
                const unsigned a1 = a0 + rs, a2 = a1 + rs;
                asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:16\n\t"
                             "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:16"
                             : "=&v"(Fn[0]), "=&v"(Fn[1]), "=&v"(Fn[2]), "=&v"(Fn[3]), "=&v"(Fn[4]), "=&v"(Fn[5])
                             : "v"(a0), "v"(a1), "v"(a2)
                             : "memory");
                if (++w_q == CH) w_q = 0;
            }
        };
        if constexpr (PRE) {
            read_factors();
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Fn[0]), "+v"(Fn[1]), "+v"(Fn[2]), "+v"(Fn[3]), "+v"(Fn[4]), "+v"(Fn[5])::"memory");
        }
        auto write_a = [&](const float4 (&L)[NPASS][2], unsigned valid, int g) {   // a loaded chunk -> split -> buffer g % NPB
            f32x4 F[PRE ? 6 : 1];
            if constexpr (PRE) {
#pragma unroll
                for (int k = 0; k < 6; ++k) F[k] = Fn[k];
            }
#pragma unroll
            for (int q = 0; q < NPASS; ++q) write_pass(L[q][0], L[q][1], valid, g, q, F);
            read_factors();                                            // (for the next chunk; waited for before this iteration's barrier)
        };
        // Iteration s (from -2; a barrier closes it from -1 on, the first one releases the consumers):
        //   A  the chunk whose first stage is s + 2 goes to LDS (its buffer was last read for a stage whose reads are
        //      complete at barrier s - 1) and the loads of the chunk NSET further on are issued into the registers it
        //      leaves;  B  the weights of stage s + AH are issued;  C  stage s + 2's weights have landed: all but the
        //      DMAs (and loads) issued after them, a constant in the periodic order.
        // Pipeline fill: the loop starts V iterations early and runs the SAME periodic order of issues -- dummies (the
        // weights' first rows into stage 0's slot, the page of zeros) where the stage or chunk an iteration would issue
        // does not exist yet -- with the same counted waits, which cannot bind before the ring is full; nothing is
        // written and no barrier is passed before iteration -2.  (The first version issued a prefix by hand and waited
        // vmcnt(0) through the first AH iterations: one full memory latency per stage, ~10 us at the start of every launch
        // -- most of the run time of the 14x14 and 7x7 layers.)
        constexpr int V = PATCH ? 9 : (NSET > AH - 2 ? NSET : AH - 2);
        constexpr int S_BEGIN = -2 - UNR * ((V + UNR - 1) / UNR);
        w_skip = -AH - S_BEGIN > 0 ? -AH - S_BEGIN : 0;
        a_skip = (!PATCH && (-2 - NSET) - S_BEGIN > 0) ? (-2 - NSET) - S_BEGIN : 0;
        float4 am4 = reinterpret_cast<const float4*>(p.amax_in)[lane];   // the tensor's amax record (used at iteration -2)
        if constexpr (PRE) {
            if (p.pre_nrec > 0) {
                const float b = pre_bound_from_table(reinterpret_cast<const float*>(ptab), p.Cin, p.amax_in, p.pre_nrec, lane);
                am4 = make_float4(b, b, b, b);
            }
        }
        int w_g = 0;                                                   // next chunk to write
#ifdef NW_CONV_DIAG
        unsigned long long cd_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cl_ = __builtin_amdgcn_s_memtime();
        const unsigned long long cf_ = cl_;
#endif
        for (int s = S_BEGIN; s < S; s += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int si = s + u;
                if (si >= S) break;
                if (si == -2) {
                    const float amx = wave_max(fmaxf(fmaxf(am4.x, am4.y), fmaxf(am4.z, am4.w)));
                    up = __builtin_ldexpf(1.f, split_exponent(amx));
                    if (PRE && !(amx < INFINITY)) up = __builtin_nanf("");   // the table saw a NaN / infinity (nw_bn_nhwc_prep_*)
                }
                if (!PATCH || u == 0) {
                    constexpr int SET_MASK = NSET - 1;
                    float4 (&L)[NPASS][2] = ld[PATCH ? 0 : (u & SET_MASK)];
                    unsigned& V_ = ldvalid[PATCH ? 0 : (u & SET_MASK)];
                    // the chunk's loads have landed: all but what was issued after them
                    NW_CSTAMP(0);                                                  // bookkeeping
                    if constexpr (PATCH) wait_vmcnt<9 * NIW>();
                    else wait_vmcnt<NSET * NIW + (NSET - 1) * NACT>();
                    NW_CSTAMP(1);                                                  // wait: the chunk's loads
                    if (si >= -2) {
                        if (w_g < QT) write_a(L, V_, w_g);
                        ++w_g;
                    }
                    NW_CSTAMP(2);                                                  // convert + LDS stores
                    issue_a(L, V_);
                    NW_CSTAMP(3);                                                  // issue of the next chunk's loads
                }
                issue_w();
                NW_CSTAMP(0);
                if constexpr (PATCH) {
                    if (u <= AH - 3) wait_vmcnt<(AH - 2) * NIW + NACT>();   // this period's chunk loads are younger too
                    else wait_vmcnt<(AH - 2) * NIW>();
                } else {
                    wait_vmcnt<(AH - 2) * (NACT + NIW)>();
                }
                if constexpr (PRE)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Fn[0]), "+v"(Fn[1]), "+v"(Fn[2]), "+v"(Fn[3]), "+v"(Fn[4]), "+v"(Fn[5])::"memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                NW_CSTAMP(4);                                                      // wait: stage s + 2's weights
                if (si >= -1) __builtin_amdgcn_s_barrier();
                NW_CSTAMP(5);                                                      // barrier
            }
        }
#ifdef NW_CONV_DIAG
        if (tid == 256 && blockIdx.x < 1024) {
            for (int k = 0; k < 6; ++k) nw_conv_diag[16 * blockIdx.x + k] = cd_[k];
            nw_conv_diag[16 * blockIdx.x + 6] = cl_ - cf_;
        }
#endif
        wait_vmcnt<0>();                                               // (the dummy tail of the issue order)
        __builtin_amdgcn_s_barrier();                                  // the consumers have parked their maxima
        return;
    }

    // ================================================================== consumers
    const int i = lane & 15, g = lane >> 4;
    const int wco = (wave % WM) * (16 * NA), wpx = (wave / WM) * (16 * NB);
    const int asw = (i >> 1) & 7;
    const int aoff_h = (wco + i) * 128 + ((g ^ asw) << 4), aoff_l = (wco + i) * 128 + (((4 + g) ^ asw) << 4);
    float inv_up = 1.f;
    if (!(PRE && p.pre_nrec > 0)) inv_up = __builtin_ldexpf(1.f, -split_exponent(amax_read(p.amax_in, lane)));
    struct Frag {
        float4 ah[NA], al[NA], bh[NB], bl[NB];
    };
    auto mm = [](const float4& a, const float4& b, f32x4 c) {
#ifdef NW_CABL_NOMFMA   // ablation builds (tools/bench_conv.hip): what the other role costs alone
        c[0] += a.x * b.x;
        return c;
#else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
#endif
    };
    float amax = 0.f;
    int sg = 0, qg = 0;                                            // global stage / chunk counters
#ifdef NW_CONV_DIAG
    unsigned long long cd_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cl_ = __builtin_amdgcn_s_memtime();
    const unsigned long long cf_ = cl_;
#define NW_CBAR() do { NW_CSTAMP(1); tile_barrier(); NW_CSTAMP(2); } while (0)
#else
#define NW_CBAR() tile_barrier()
#endif
    // moments of everything this wave row has written so far (p_macc): Chan's merge, tile by tile, in registers (MACC_LDS: in LDS)
    float rcnt = 0.f, rmean[NA][4], rm2[NA][4], rlo[NA][4], rhi[NA][4];
    float4* const mlds = reinterpret_cast<float4*>(smem + C::LDS) + (wave * 4 + g) * (4 * NA);   // [a]: mean, m2, min, max
#pragma unroll
    for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { rmean[a][e] = rm2[a][e] = 0.f; rlo[a][e] = INFINITY; rhi[a][e] = -INFINITY; }
        if (MACC_LDS && STATS && i == 0) {
            mlds[4 * a] = mlds[4 * a + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
            mlds[4 * a + 2] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
            mlds[4 * a + 3] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
    }
    auto run_mm = [&](int a, const float (&lo)[4], const float (&hi)[4]) {   // running minimum / maximum of the workgroup's rows
        if (MACC_LDS) {
            const float4 x = mlds[4 * a + 2], y = mlds[4 * a + 3];
            mlds[4 * a + 2] = make_float4(fminf(x.x, lo[0]), fminf(x.y, lo[1]), fminf(x.z, lo[2]), fminf(x.w, lo[3]));
            mlds[4 * a + 3] = make_float4(fmaxf(y.x, hi[0]), fmaxf(y.y, hi[1]), fmaxf(y.z, hi[2]), fmaxf(y.w, hi[3]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { rlo[a][e] = fminf(rlo[a][e], lo[e]); rhi[a][e] = fmaxf(rhi[a][e], hi[e]); }
        }
    };
    auto run_get_mm = [&](int a, float (&lo)[4], float (&hi)[4]) {
        if (MACC_LDS) {
            const float4 x = mlds[4 * a + 2], y = mlds[4 * a + 3];
            lo[0] = x.x; lo[1] = x.y; lo[2] = x.z; lo[3] = x.w; hi[0] = y.x; hi[1] = y.y; hi[2] = y.z; hi[3] = y.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = rlo[a][e]; hi[e] = rhi[a][e]; }
        }
    };
    auto run_get = [&](int a, float (&m)[4], float (&q)[4]) {
        if (MACC_LDS) {
            const float4 x = mlds[4 * a], y = mlds[4 * a + 1];
            m[0] = x.x; m[1] = x.y; m[2] = x.z; m[3] = x.w; q[0] = y.x; q[1] = y.y; q[2] = y.z; q[3] = y.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { m[e] = rmean[a][e]; q[e] = rm2[a][e]; }
        }
    };
    auto run_put = [&](int a, const float (&m)[4], const float (&q)[4]) {
        if (MACC_LDS) {
            mlds[4 * a] = make_float4(m[0], m[1], m[2], m[3]);
            mlds[4 * a + 1] = make_float4(q[0], q[1], q[2], q[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { rmean[a][e] = m[e]; rm2[a][e] = q[e]; }
        }
    };
    if constexpr (PRE) {
        __builtin_amdgcn_s_barrier();                              // (the loaders have staged the BatchNorm table)
        if (p.pre_nrec > 0)
            inv_up = __builtin_ldexpf(1.f, -split_exponent(pre_bound_from_table(reinterpret_cast<const float*>(ptab), p.Cin, p.amax_in,
                                                                                 p.pre_nrec, lane)));
    }
    __builtin_amdgcn_s_barrier();                                  // the prologue's stages have landed
    NW_CSTAMP(0);                                                  // wait for the pipeline fill
    for (int tl = 0; tl < ntile; ++tl) {
        const int t_id = tbeg + tl * nslot;
        const int mt = t_id / p.ntiles, nt = t_id - mt * p.ntiles;
        const int m0 = mt * BM, co0 = nt * BN;
        int eb[NB];
        if (PATCH) {
            const int n0 = m0 / HW, r0 = m0 - n0 * HW, y0 = r0 / p.Wo, x0 = r0 - y0 * p.Wo;
            const int lo = n0 * p.IMG + y0 * p.IP + x0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int m = min(m0 + wpx + 16 * b + i, p.M - 1);
                const int n = m / HW, r = m - n * HW, yo = r / p.Wo, xo = r - yo * p.Wo;
                eb[b] = n * p.IMG + yo * p.IP + xo - lo;
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) eb[b] = wpx + 16 * b + i;
        }
        f32x4 acc[NA][NB];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto load_frags = [&](Frag& f, int k) {                    // stage k of this tile
            const char* wb = wring + ((sg + k) % NWR) * WST;
            int shift = 0, q = k;
            if (PATCH) {
                q = k / 9;
                const int t = k - 9 * q, dy = t / 3, dx = t - 3 * dy;
                shift = dy * p.IP + dx;
            }
            const char* pb = pbuf + ((qg + q) % NPB) * PB;
#ifdef NW_CABL_NOFRAG   // ablation: fragments read once (first stage of the kernel), the rest of the stages reuse the registers
            if (sg + k > 0) return;
#endif
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                f.ah[a] = *reinterpret_cast<const float4*>(wb + aoff_h + a * 2048);
                f.al[a] = *reinterpret_cast<const float4*>(wb + aoff_l + a * 2048);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int e = eb[b] + shift;
                f.bh[b] = *reinterpret_cast<const float4*>(pb + e * 128 + (pslot(e, g) << 4));
                f.bl[b] = *reinterpret_cast<const float4*>(pb + e * 128 + (pslot(e, 4 + g) << 4));
            }
        };
        auto mfma_stage = [&](const Frag& f) {                     // small terms first, the dominant h*h product last
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = mm(f.al[a], f.bh[b], acc[a][b]);
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = mm(f.ah[a], f.bl[b], acc[a][b]);
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = mm(f.ah[a], f.bh[b], acc[a][b]);
        };
        auto interleave = [&]() {
            constexpr int RD = 2 * (NA + NB), MF = 3 * NA * NB;
            constexpr int PAIR = RD < MF ? RD : MF;
#pragma unroll
            for (int x = 0; x < PAIR; ++x) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (MF > PAIR) __builtin_amdgcn_sched_group_barrier(0x008, MF - PAIR, 0);
        };
        // the epilogue's per-channel factors are requested at the top of the tile's LAST stage: their L2 latency (~1 us, once per
        // tile, on the critical path when they were loaded where they are used) passes behind that stage's MFMAs, in registers
        // the second fragment set no longer needs (requested at tile set-up they were live across the whole main loop: the
        // 128 x 128 tiles spilled)
        float4 ws4[NA], b4[NA];
        Frag f0, f1;
        load_frags(f0, 0);
        NW_CSTAMP(3);                                              // tile set-up (+ the first fragments' issue)
        int k = 0;
        for (; k + 2 < ST; k += 2) {
            load_frags(f1, k + 1);
            mfma_stage(f0);
            interleave();
            NW_CBAR();
            load_frags(f0, k + 2);
            mfma_stage(f1);
            interleave();
            NW_CBAR();
        }
        auto load_factors = [&]() {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int co = co0 + wco + 16 * a + 4 * g;
                ws4[a] = *reinterpret_cast<const float4*>(p.wscale + co);
                b4[a] = p_bias ? *reinterpret_cast<const float4*>(p_bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        if (k + 1 < ST) {                                          // two stages left: the factors ride in f0's registers
            load_frags(f1, k + 1);
            mfma_stage(f0);
            NW_CBAR();
            load_factors();
            mfma_stage(f1);
            NW_CBAR();
        } else {                                                   // one stage left: ... in f1's
            load_factors();
            mfma_stage(f0);
            NW_CBAR();
        }
        sg += ST;
        qg += CH;
        // ---- store: acc[a][b][e] of lane (i, g) = y[pixel m0 + wpx + 16 b + i][channel co0 + wco + 16 a + 4 g + e]
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int co = co0 + wco + 16 * a + 4 * g;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int m = m0 + wpx + 16 * b + i;
                if (m < p.M) {
                    float4 v;
                    v.x = __builtin_fmaf(acc[a][b][0], ws4[a].x * inv_up, b4[a].x);
                    v.y = __builtin_fmaf(acc[a][b][1], ws4[a].y * inv_up, b4[a].y);
                    v.z = __builtin_fmaf(acc[a][b][2], ws4[a].z * inv_up, b4[a].z);
                    v.w = __builtin_fmaf(acc[a][b][3], ws4[a].w * inv_up, b4[a].w);
                    const size_t o = (size_t)m * p.ldy + co;
                    if (p_res) {
                        const float4 r = *reinterpret_cast<const float4*>(p_res + (size_t)m * p.Cout + co);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    if (p_relu) {   // (x < 0 ? 0 : x keeps a NaN, like torch's relu)
                        v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y;
                        v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
                    }
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
#ifdef NW_CABL_NOEPI
                    if (v.x == 12345.678f)
#endif
                    *reinterpret_cast<float4*>(p.y + o) = v;
                    acc[a][b] = f32x4{v.x, v.y, v.z, v.w};
                }
            }
        }
        if (p_moments) {
            // Welford moments of this wave's pixel rows [m0 + wpx, + 16 NB) per output channel, two passes over the values
            // still in registers (mean, then centred squares); the 16 lanes of a row hold 16 pixels of the same 4 channels:
            // summed by four DPP rotations.  One group per (tile row, wave row); merged with Chan's formula by the finalize
            // kernel of bn_nhwc.hip in a fixed order.
            const int nw_ = min(max(p.M - (m0 + wpx), 0), 16 * NB);
            const float cnt = (float)nw_, rc = nw_ > 0 ? 1.f / cnt : 0.f;
            auto rowsum = [](float x) {
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));  // row_ror:8
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));  // row_ror:4
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
                return x;
            };
            const int G = p.mtiles * WN, grp = mt * WN + wave / WM;
            const float ntot = rcnt + cnt, wt = ntot > 0.f ? cnt / ntot : 0.f;
            auto rowmin = [](float x) {
                x = fminf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false)));
                x = fminf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false)));
                x = fminf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4e, 0xf, 0xf, false)));
                x = fminf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xb1, 0xf, 0xf, false)));
                return x;
            };
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                float mean[4], m2[4], lo[4], hi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = 0.f, vlo = INFINITY, vhi = -INFINITY;
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const bool ok = m0 + wpx + 16 * b + i < p.M;
                        s += ok ? acc[a][b][e] : 0.f;
                        vlo = fminf(vlo, ok ? acc[a][b][e] : INFINITY);
                        vhi = fmaxf(vhi, ok ? acc[a][b][e] : -INFINITY);
                    }
                    lo[e] = rowmin(vlo);
                    hi[e] = -rowmin(-vhi);
                    mean[e] = rowsum(s) * rc;
                    float q = 0.f;
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float d = (m0 + wpx + 16 * b + i < p.M) ? acc[a][b][e] - mean[e] : 0.f;
                        q = __builtin_fmaf(d, d, q);
                    }
                    m2[e] = rowsum(q);
                }
                if (p_macc) {
                    if (!MACC_LDS || i == 0) {
                        float rm[4], rq[4];
                        run_get(a, rm, rq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d = mean[e] - rm[e];
                            rm[e] = __builtin_fmaf(d, wt, rm[e]);
                            rq[e] += __builtin_fmaf(d * d, rcnt * wt, m2[e]);
                        }
                        run_put(a, rm, rq);
                        run_mm(a, lo, hi);
                    }
                } else if (i == 0) {
                    const int co = co0 + wco + 16 * a + 4 * g;
                    *reinterpret_cast<float4*>(p_moments + ((size_t)3 * G + grp) * p.Cout + co) = make_float4(lo[0], lo[1], lo[2], lo[3]);
                    *reinterpret_cast<float4*>(p_moments + ((size_t)4 * G + grp) * p.Cout + co) = make_float4(hi[0], hi[1], hi[2], hi[3]);
                    *reinterpret_cast<float4*>(p_moments + ((size_t)0 * G + grp) * p.Cout + co) = make_float4(cnt, cnt, cnt, cnt);
                    *reinterpret_cast<float4*>(p_moments + ((size_t)1 * G + grp) * p.Cout + co) = make_float4(mean[0], mean[1], mean[2], mean[3]);
                    *reinterpret_cast<float4*>(p_moments + ((size_t)2 * G + grp) * p.Cout + co) = make_float4(m2[0], m2[1], m2[2], m2[3]);
                }
            }
            if (p_macc) rcnt = ntot;
        }
        if (p_bnb_part) {
            // y = dL/d relu(bn(x)): sum g and sum g xhat over this wave's pixel rows, g = y where the forward's own
            // bn(x) = (x - mean) (gamma invstd) + beta was positive (the same expression nw_bn_nhwc_bwd_stats_kernel uses)
            auto rowsum = [](float x) {
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4e, 0xf, 0xf, false));
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xb1, 0xf, 0xf, false));
                return x;
            };
            const int G = p.mtiles * WN, grp = mt * WN + wave / WM;
            // every x value first (rows past the end clamped: their g is zeroed below), so that the loads fly together.  (Requested
            // in front of the store loop instead, their registers stay live across it in EVERY instantiation: all convolutions
            // 4-5 % slower, K2 1.427 -> 1.494 ms.)
            float4 x4[NA][NB];
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int m = min(m0 + wpx + 16 * b + i, p.M - 1);
                    x4[a][b] = *reinterpret_cast<const float4*>(p.bnb_x + (size_t)m * p.bnb_ldx + (co0 + wco + 16 * a + 4 * g));
                }
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int co = co0 + wco + 16 * a + 4 * g;
                const float4 m4 = *reinterpret_cast<const float4*>(p.bnb_mean + co), i4 = *reinterpret_cast<const float4*>(p.bnb_invstd + co),
                             g4 = *reinterpret_cast<const float4*>(p.bnb_gamma + co), b4 = *reinterpret_cast<const float4*>(p.bnb_beta + co);
                const float mean[4] = {m4.x, m4.y, m4.z, m4.w}, inv[4] = {i4.x, i4.y, i4.z, i4.w};
                const float sa[4] = {g4.x * i4.x, g4.y * i4.y, g4.z * i4.z, g4.w * i4.w}, sb[4] = {b4.x, b4.y, b4.z, b4.w};
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const bool ok = m0 + wpx + 16 * b + i < p.M;
                    const float xv[4] = {x4[a][b].x, x4[a][b].y, x4[a][b].z, x4[a][b].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xm = xv[e] - mean[e];
                        const float pre = __builtin_fmaf(xm, sa[e], sb[e]);
                        const float gd = (ok && pre > 0.f) ? acc[a][b][e] : 0.f;
                        s1[e] += gd;
                        s2[e] = __builtin_fmaf(gd, xm * inv[e], s2[e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] = rowsum(s1[e]); s2[e] = rowsum(s2[e]); }
                if (p_macc) {                                      // one group per workgroup and wave row (the running moments'
                    if (!MACC_LDS || i == 0) {                     //  storage is free: the two have separate entry points)
                        float rm[4], rq[4];
                        run_get(a, rm, rq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { rm[e] += s1[e]; rq[e] += s2[e]; }
                        run_put(a, rm, rq);
                    }
                } else if (i == 0) {
                    *reinterpret_cast<float4*>(p_bnb_part + ((size_t)0 * G + grp) * p.Cout + co) = make_float4(s1[0], s1[1], s1[2], s1[3]);
                    *reinterpret_cast<float4*>(p_bnb_part + ((size_t)1 * G + grp) * p.Cout + co) = make_float4(s2[0], s2[1], s2[2], s2[3]);
                }
            }
        }
        NW_CSTAMP(4);                                              // epilogue of the tile
    }
#ifdef NW_CONV_DIAG
    if (tid == 0 && blockIdx.x < 1024) {
        for (int k = 0; k < 5; ++k) nw_conv_diag[16 * blockIdx.x + 8 + k] = cd_[k];
        nw_conv_diag[16 * blockIdx.x + 14] = cl_ - cf_;
    }
#endif
    if (p_bnb_part && p_macc && i == 0) {
        const int G = gridDim.x * WN, grp = blockIdx.x * WN + wave / WM;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int co = wco + 16 * a + 4 * g;
            float rm[4], rq[4];
            run_get(a, rm, rq);
            *reinterpret_cast<float4*>(p_bnb_part + ((size_t)0 * G + grp) * p.Cout + co) = make_float4(rm[0], rm[1], rm[2], rm[3]);
            *reinterpret_cast<float4*>(p_bnb_part + ((size_t)1 * G + grp) * p.Cout + co) = make_float4(rq[0], rq[1], rq[2], rq[3]);
        }
    }
    if (p_moments && p_macc && i == 0) {
        const int G = gridDim.x * WN, grp = blockIdx.x * WN + wave / WM;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int co = wco + 16 * a + 4 * g;                   // (ntiles == 1: the tile starts at channel 0)
            *reinterpret_cast<float4*>(p_moments + ((size_t)0 * G + grp) * p.Cout + co) = make_float4(rcnt, rcnt, rcnt, rcnt);
            float rm[4], rq[4];
            run_get(a, rm, rq);
            *reinterpret_cast<float4*>(p_moments + ((size_t)1 * G + grp) * p.Cout + co) = make_float4(rm[0], rm[1], rm[2], rm[3]);
            *reinterpret_cast<float4*>(p_moments + ((size_t)2 * G + grp) * p.Cout + co) = make_float4(rq[0], rq[1], rq[2], rq[3]);
            float lo[4], hi[4];
            run_get_mm(a, lo, hi);
            *reinterpret_cast<float4*>(p_moments + ((size_t)3 * G + grp) * p.Cout + co) = make_float4(lo[0], lo[1], lo[2], lo[3]);
            *reinterpret_cast<float4*>(p_moments + ((size_t)4 * G + grp) * p.Cout + co) = make_float4(hi[0], hi[1], hi[2], hi[3]);
        }
    }
    // this workgroup's maximum -> its slot of the output's amax record (no atomics, nothing to clear beforehand)
    amax = wave_max(amax);
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) red[wave] = amax;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.amax_out && tid == 0)
        amax_write(p.amax_out, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), blockIdx.x, gridDim.x);
}

// max |x| over a dense fp32 array as an amax record: block b's maximum in slot b
__global__ __launch_bounds__(256) void nw_absmax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float red[8];
    float m = 0.f;
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (int64_t)gridDim.x * 256) {
        const float4 v = x4[k];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int64_t k = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        m = fmaxf(m, fabsf(x[k]));
    m = block_max(m, red);
    if (threadIdx.x == 0) amax_write(out, m, blockIdx.x, gridDim.x);
}

// (n, c, hw) fp32 with strides (sn, sc, sp) -> (n, hw, cp) channels-last with channels c .. cp-1 zero, and max |x|:
// the network input (NCHW from the loader, or channels_last) becomes the 4-channel NHWC tensor the stem reads
// (at most CV_AMAX_SLOTS workgroups -- one amax slot each --, so 1024 lanes per workgroup: with 256 the 3.2 M pixels of a 64 x
//  224 x 224 batch were 49 dependent iterations per lane, 43 us for 90 MB)
__global__ __launch_bounds__(1024) void nw_to_nhwc_pad_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               float* __restrict__ amax, int64_t npix, int c, int hw,
                                                               int cp, int64_t sn, int64_t sc, int64_t sp) {
    __shared__ float red[16];
    float m = 0.f;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = pix / hw, p = pix - n * hw;
        const float* src = x + n * sn + p * sp;
        float* dst = y + pix * cp;
        for (int c0 = 0; c0 < cp; c0 += 4) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = c0 + j < c ? src[(c0 + j) * sc] : 0.f;
                m = fmaxf(m, fabsf(v[j]));
            }
            *reinterpret_cast<float4*>(dst + c0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) amax_write(amax, m, blockIdx.x, gridDim.x);
}

const float4* zero_page() {   // device address of nw_conv_zeros on the current device
    static const float4* cache[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!cache[dev]) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(nw_conv_zeros)) != hipSuccess) return nullptr;
        cache[dev] = static_cast<const float4*>(ptr);
    }
    return cache[dev];
}

int num_cus() {
    static const int v = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return v;
}

// moments_groups (host, nullable): receives the number of groups the moments buffer holds (mtiles x wave rows);
// dry: only report it (nw_conv2d_nhwc_moments_groups)
template <int NA, int NB, int WM, int MODE>
int launch_conv_cfg(ConvP p, hipStream_t st, int64_t* moments_groups = nullptr, bool dry = false) {
    using C = ConvCfg<NA, NB, WM, MODE>;
    p.mtiles = (p.M + C::BM - 1) / C::BM;
    p.ntiles = p.Cout / C::BN;
    const int64_t total = (int64_t)p.mtiles * p.ntiles;
    int64_t grid = num_cus() < CV_AMAX_SLOTS ? num_cus() : CV_AMAX_SLOTS;
    const int cap = knob(KNOB_CONV_MAX_WGS);   // tests: many tiles per workgroup (diagnostic knob "conv_max_wgs")
    if (cap > 0 && grid > cap) grid = cap;
    if (grid > total) grid = total;
    grid = (grid + 7) / 8 * 8;
    // one output-channel tile: a workgroup's tiles are all rows of the same channels, and their moments are merged in its
    // registers -- grid x WN groups for the merge kernel instead of mtiles x WN (2058 -> 512 on the 56 x 56 layers)
    p.macc = p.ntiles == 1 && knob(KNOB_CONV_MOMENTS_PER_TILE) <= 0;
    if (moments_groups) *moments_groups = (p.macc ? grid : (int64_t)p.mtiles) * C::WN;
    if (dry) return NW_OK;
    constexpr size_t lds = C::LDS + (NA == 4 && NB == 4 ? 4096 : 0);   // (+ the 128 x 128 tile's running moments, minima, maxima)
    static_assert(lds <= 160 * 1024, "LDS");
    // three instantiations: plain (training: data gradients, transitions), with statistics (training forward), with the
    // inference epilogue (bias / identity / ReLU)
    auto kern0 = nw_conv_nhwc_kernel<NA, NB, WM, MODE, false, false>;
    auto kern1 = nw_conv_nhwc_kernel<NA, NB, WM, MODE, true, false>;
    auto kern2 = nw_conv_nhwc_kernel<NA, NB, WM, MODE, false, true>;
    auto kern3 = nw_conv_nhwc_kernel<NA, NB, WM, MODE, true, true>;
    static const bool attr = [&] {
        bool ok = true;
        for (const void* k : {reinterpret_cast<const void*>(kern0), reinterpret_cast<const void*>(kern1),
                              reinterpret_cast<const void*>(kern2), reinterpret_cast<const void*>(kern3)})
            ok = ok && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        return ok;
    }();
    if (!attr) return NW_ERR_LAUNCH;
    const bool stats = p.moments || p.bnb_part, post = p.bias || p.res || p.relu;
    if (p.pre) {   // BatchNorm + ReLU in the loaders: training forward (statistics), inference (bias / ReLU behind it) or plain
        if constexpr (MODE == CV_GATHER || MODE == CV_PATCH) {
            using CP = ConvCfg<NA, NB, WM, MODE, true>;
            constexpr size_t ldsp0 = CP::LDS + (NA == 4 && NB == 4 ? 4096 : 0);
            static_assert(ldsp0 <= 160 * 1024, "LDS");
            const size_t ldsp = ldsp0 + (size_t)12 * p.Cin;        // + the BatchNorm table of the input channels
            if (ldsp > 160 * 1024) return NW_ERR_UNSUPPORTED;
            auto kp0 = nw_conv_nhwc_kernel<NA, NB, WM, MODE, false, false, true>;
            auto kp1 = nw_conv_nhwc_kernel<NA, NB, WM, MODE, true, false, true>;
            auto kp2 = nw_conv_nhwc_kernel<NA, NB, WM, MODE, false, true, true>;
            static const bool attrp = [&] {
                bool ok = true;
                for (const void* k : {reinterpret_cast<const void*>(kp0), reinterpret_cast<const void*>(kp1), reinterpret_cast<const void*>(kp2)})
                    ok = ok && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
                return ok;
            }();
            if (!attrp) return NW_ERR_LAUNCH;
            if (stats && post) return NW_ERR_UNSUPPORTED;
            auto kp = stats ? kp1 : (post ? kp2 : kp0);
            hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(512), ldsp, st, p);
            NW_CHECK_LAUNCH();
            return NW_OK;
        } else {
            return NW_ERR_UNSUPPORTED;
        }
    }
    auto kern = stats ? (post ? kern3 : kern1) : (post ? kern2 : kern0);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, p);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

}  // namespace
}  // namespace nw

extern "C" const void* nw_conv_zero_page(void) { return nw::zero_page(); }

extern "C" int nw_absmax_f32(const float* x, int64_t count, float* amax_out, void* stream) {
    if (count < 0 || !amax_out || (count > 0 && !x)) return NW_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(amax_out)) & 15) return NW_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t blocks = (count / 4 + 255) / 256;
    if (blocks > nw::CV_AMAX_SLOTS) blocks = nw::CV_AMAX_SLOTS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(nw::nw_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, count, amax_out);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_to_nhwc_pad_f32(const float* x, float* y, float* amax_out, int64_t n, int64_t c, int64_t hw, int64_t cp,
                                  int64_t stride_n, int64_t stride_c, int64_t stride_p, void* stream) {
    if (n < 0 || c <= 0 || hw < 0 || cp < c || cp % 4 || !amax_out) return NW_ERR_INVALID_ARG;
    if ((n * hw > 0 && (!x || !y)) || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(amax_out)) & 15)) return NW_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t blocks = (n * hw + 1023) / 1024;
    if (blocks > nw::CV_AMAX_SLOTS) blocks = nw::CV_AMAX_SLOTS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(nw::nw_to_nhwc_pad_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, x, y, amax_out, n * hw, (int)c,
                       (int)hw, (int)cp, stride_n, stride_c, stride_p);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

extern "C" int nw_conv2d_nhwc_supported(int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                                        int64_t stride, int64_t pad) {
    if (n <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
    if (Cout % 32 || KH > 15 || KW > 15) return 0;
    if (Cin % 32 && Cin * KW > 32) return 0;   // (few input channels: one kernel row = one 32-wide k chunk)
    const int64_t Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    // 32-bit element offsets inside the kernels
    if (n * H * W * Cin >= (1LL << 31) || n * Ho * Wo * Cout >= (1LL << 31) || Cout * KH * KW * Cin >= (1LL << 29)) return 0;
    if (n * (H + pad) * (W + pad) >= (1LL << 30)) return 0;
    return 1;
}

static int conv2d_nhwc_impl(const float* x, const float* amax_in, const float* w_split, const float* w_scale,
                           const float* bias, const float* residual, int relu, float* y, float* amax_out,
                           int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                           int64_t stride, int64_t pad, int64_t ldx, int64_t ldy, float* moments, int64_t* moments_groups,
                           bool dry, void* stream, const nw_conv_bnstat* bnstat = nullptr, const float* pre = nullptr,
                           int64_t pre_nrec = 0) {
    if (n < 0 || H < 0 || W < 0) return NW_ERR_INVALID_ARG;
    if (moments_groups) *moments_groups = 0;
    if (n == 0) return NW_OK;
    if (!nw_conv2d_nhwc_supported(n, H, W, Cin, Cout, KH, KW, stride, pad)) return NW_ERR_UNSUPPORTED;
    if (ldx == 0) ldx = Cin;
    if (ldy == 0) ldy = Cout;
    if (ldx < Cin || ldy < Cout || (ldx != Cin && ldx % 4) || ldy % 4) return NW_ERR_INVALID_ARG;
    if (ldx != Cin && Cin % 32) return NW_ERR_UNSUPPORTED;   // (the few-channel stems read whole dense rows)
    {
        const int64_t Ho_ = (H + 2 * pad - KH) / stride + 1, Wo_ = (W + 2 * pad - KW) / stride + 1;
        if (n * H * W * ldx >= (1LL << 31) || n * Ho_ * Wo_ * ldy >= (1LL << 31)) return NW_ERR_UNSUPPORTED;
    }
    if (!dry && (!x || !amax_in || !w_split || !w_scale || !y)) return NW_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_split) | reinterpret_cast<uintptr_t>(y) |
         reinterpret_cast<uintptr_t>(w_scale) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual) |
         reinterpret_cast<uintptr_t>(amax_in) | reinterpret_cast<uintptr_t>(amax_out) | reinterpret_cast<uintptr_t>(moments)) & 15)
        return NW_ERR_INVALID_ARG;
    if ((moments || dry) && Cin % 32) return NW_ERR_UNSUPPORTED;   // (the few-channel stems leave no moments)
    hipStream_t st = static_cast<hipStream_t>(stream);
    nw::ConvP p;
    p.x = x; p.amax_in = amax_in; p.ws = reinterpret_cast<const char*>(w_split); p.wscale = w_scale; p.bias = bias;
    p.res = residual; p.y = y; p.amax_out = amax_out; p.moments = moments;
    p.pre = pre;
    p.pre_nrec = (int)pre_nrec;
    if (pre_nrec < 0 || pre_nrec > 4096 || (!pre && pre_nrec)) return NW_ERR_INVALID_ARG;
    if (pre && (Cin % 32 || (reinterpret_cast<uintptr_t>(pre) & 15))) return NW_ERR_INVALID_ARG;
    p.bnb_x = p.bnb_mean = p.bnb_invstd = p.bnb_gamma = p.bnb_beta = nullptr; p.bnb_part = nullptr; p.bnb_ldx = 0;
    if (bnstat) {
        if (Cin % 32) return NW_ERR_UNSUPPORTED;
        if (!bnstat->x || !bnstat->mean || !bnstat->invstd || !bnstat->gamma || !bnstat->beta || !bnstat->partials ||
            bnstat->ldx < Cout || bnstat->ldx % 4 || n * ((H + 2 * pad - KH) / stride + 1) * ((W + 2 * pad - KW) / stride + 1) * bnstat->ldx >= (1LL << 31))
            return NW_ERR_INVALID_ARG;
        if ((reinterpret_cast<uintptr_t>(bnstat->x) | reinterpret_cast<uintptr_t>(bnstat->mean) | reinterpret_cast<uintptr_t>(bnstat->invstd) |
             reinterpret_cast<uintptr_t>(bnstat->gamma) | reinterpret_cast<uintptr_t>(bnstat->beta) | reinterpret_cast<uintptr_t>(bnstat->partials)) & 15)
            return NW_ERR_INVALID_ARG;
        p.bnb_x = bnstat->x; p.bnb_mean = bnstat->mean; p.bnb_invstd = bnstat->invstd; p.bnb_gamma = bnstat->gamma;
        p.bnb_beta = bnstat->beta; p.bnb_part = bnstat->partials; p.bnb_ldx = (int)bnstat->ldx;
    }
    p.zeros = dry ? nullptr : nw::zero_page();
    if (!dry && !p.zeros) return NW_ERR_LAUNCH;
    p.N = (int)n; p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Cout = (int)Cout; p.KH = (int)KH; p.KW = (int)KW;
    p.stride = (int)stride; p.pad = (int)pad; p.relu = relu;
    p.ldx = (int)ldx; p.ldy = (int)ldy;
    p.Ho = (int)((H + 2 * pad - KH) / stride + 1);
    p.Wo = (int)((W + 2 * pad - KW) / stride + 1);
    p.IP = (int)(W + pad);
    p.IMG = (int)((H + pad) * (W + pad));
    p.M = (int)(n * p.Ho * p.Wo);
    p.mtiles = p.ntiles = 0;
    // PATCH mode: 3x3 / stride 1 / padding 1 whose patch (the tile's pixels, one raster row and one pixel on either side,
    // the gaps of the row and image boundaries the tile crosses) fits the LDS buffer
    const bool k33 = KH == 3 && KW == 3 && stride == 1 && pad == 1;
    auto patch_fits = [&](int BM) {
        const int64_t rc = (BM - 1 + W - 1) / W, ic = (BM - 1 + H * W - 1) / (H * W);
        return BM + rc * (p.IP - W) + ic * pad * p.IP + 2 * p.IP + 2 <= BM + 192;
    };
    const int force_gather = nw::knob(nw::KNOB_CONV_GATHER) == 1;
    if (Cin % 32) {   // ROWRUN: w_split is the split form of the (Cout, KH, 32) matrix [co][ky][kx * Cin + ci], zero-padded
        if (Cin == 4 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
            if (Cout % 64 == 0) return nw::launch_conv_cfg<4, 2, 1, nw::CV_ROWRUN4>(p, st);
            return nw::launch_conv_cfg<2, 4, 1, nw::CV_ROWRUN4>(p, st);
        }
        if (Cout % 64 == 0) return nw::launch_conv_cfg<4, 2, 1, nw::CV_ROWRUN>(p, st);
        return nw::launch_conv_cfg<2, 4, 1, nw::CV_ROWRUN>(p, st);
    }
    // The largest tile that still gives ~3/4 of the CUs a tile, else the smallest (the 14x14 and 7x7 layers are
    // latency-bound: 42 x 196 pixels are 33 tiles of 256 pixels, or 129 of 64)
    const int64_t want = (int64_t)nw::num_cus() * 3 / 4;
    auto tiles = [&](int bm, int bn) { return ((int64_t)p.M + bm - 1) / bm * (Cout / bn); };
    const int skip = nw::knob(nw::KNOB_CONV_SKIP_CFGS) > 0 ? nw::knob(nw::KNOB_CONV_SKIP_CFGS) : 0;   // timing experiments: pass over the first n fitting shapes
    const int force = nw::knob(nw::KNOB_CONV_FORCE_CFG) > 0 ? nw::knob(nw::KNOB_CONV_FORCE_CFG) : 0;   // ... or run shape number n of the list below
    int seen = 0, idx = 0;
#define NW_CONV_TRY(NA_, NB_, WM_, BM_, BN_, COND_)                                                            \
    ++idx;                                                                                                      \
    if (Cout % BN_ == 0 && (force ? idx == force : ((COND_) && seen++ >= skip))) {                              \
        if (k33 && patch_fits(BM_) && !force_gather)                                                            \
            return nw::launch_conv_cfg<NA_, NB_, WM_, nw::CV_PATCH>(p, st, moments_groups, dry);                \
        return nw::launch_conv_cfg<NA_, NB_, WM_, nw::CV_GATHER>(p, st, moments_groups, dry);                   \
    }
    const int64_t G = nw::num_cus();
    // 64 pixels x 128 channels: where the 64 x 64 tiles just miss one round of the grid (42 x 14 x 14 pixels x 128 channels:
    // 258 tiles on 256 workgroups -- two of them run twice as long as the rest) and this shape fits in one
    const bool wide = nw::knob(nw::KNOB_CONV_FORCE_CFG) != -1 && tiles(64, 64) > G && tiles(64, 64) <= G + G / 4 && tiles(64, 128) >= G / 4;
    NW_CONV_TRY(4, 4, 2, 128, 128, tiles(128, 128) >= want)
    NW_CONV_TRY(4, 2, 1, 128, 64, tiles(128, 64) >= want)
    NW_CONV_TRY(4, 2, 2, 64, 128, wide)
    if (Cout % 64 == 0) { NW_CONV_TRY(4, 1, 1, 64, 64, true) } else { ++idx; }
    NW_CONV_TRY(2, 4, 1, 256, 32, tiles(256, 32) >= want)
    NW_CONV_TRY(2, 2, 1, 128, 32, tiles(128, 32) >= want)
    NW_CONV_TRY(2, 1, 1, 64, 32, true)
#undef NW_CONV_TRY
    return NW_ERR_UNSUPPORTED;
}

extern "C" int nw_conv2d_nhwc_f16x2(const float* x, const float* amax_in, const float* w_split, const float* w_scale,
                                    const float* bias, const float* residual, int relu, float* y, float* amax_out,
                                    int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                                    int64_t stride, int64_t pad, int64_t ldx, int64_t ldy, float* moments, void* stream) {
    int64_t groups = 0;
    return conv2d_nhwc_impl(x, amax_in, w_split, w_scale, bias, residual, relu, y, amax_out, n, H, W, Cin, Cout, KH, KW, stride,
                            pad, ldx, ldy, moments, moments ? &groups : nullptr, false, stream);
}

extern "C" int nw_conv2d_nhwc_bnrelu_f16x2(const float* x, const float* pre, const float* amax_in, int64_t raw_records,
                                           const float* w_split, const float* w_scale, const float* bias, int relu, float* y,
                                           float* amax_out, int64_t n,
                                           int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
                                           int64_t pad, int64_t ldx, int64_t ldy, float* moments, void* stream) {
    if (!pre) return NW_ERR_INVALID_ARG;
    int64_t groups = 0;
    return conv2d_nhwc_impl(x, amax_in, w_split, w_scale, bias, nullptr, relu, y, amax_out, n, H, W, Cin, Cout, KH, KW, stride, pad,
                            ldx, ldy, moments, moments ? &groups : nullptr, false, stream, nullptr, pre, raw_records);
}

extern "C" int nw_conv2d_nhwc_bnstat_f16x2(const float* x, const float* amax_in, const float* w_split, const float* w_scale, float* y,
                                           float* amax_out, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH,
                                           int64_t KW, int64_t stride, int64_t pad, int64_t ldx, int64_t ldy,
                                           const nw_conv_bnstat* bnstat, void* stream) {
    if (!bnstat) return NW_ERR_INVALID_ARG;
    int64_t groups = 0;
    return conv2d_nhwc_impl(x, amax_in, w_split, w_scale, nullptr, nullptr, 0, y, amax_out, n, H, W, Cin, Cout, KH, KW, stride, pad,
                            ldx, ldy, nullptr, &groups, false, stream, bnstat);
}

extern "C" int64_t nw_conv2d_nhwc_moments_groups(int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH,
                                                 int64_t KW, int64_t stride, int64_t pad) {
    int64_t groups = 0;
    if (n <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;   // (validated like the call itself)
    const int rc = conv2d_nhwc_impl(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, n, H, W, Cin, Cout,
                                    KH, KW, stride, pad, 0, 0, nullptr, &groups, true, nullptr);
    return rc == NW_OK ? groups : 0;
}
