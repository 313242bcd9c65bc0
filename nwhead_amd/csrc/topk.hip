// topk.hip -- the k best supports of every query from a (B,N) score matrix (gfx950 / MI355X only).
//
// Replaces the full descending argsort the reference runs for its neighbour modes and then cuts to k
// columns (KNN.__call__, nwhead/utils.py:185-193; NWNet.get_neighbors, nwhead/nw.py:245-249):
//   idx[b][0..k) = the first k columns of argsort(scores[b], descending, stable)
// i.e. best score first, equal scores in ascending index order -- bit-exact, not approximate.
//
// One 1024-thread workgroup per row (four independent loads in flight per thread in the histogram passes):
//   1. radix select on the order-preserving uint image of the floats, four 8-bit digits from the top:
//      an LDS histogram of the digit among the elements that match the prefix found so far gives the
//      digit of the k-th largest; lanes of a wave that hit the same bin are merged before the LDS
//      atomic (the top byte of a score row is almost constant: one hot bin);
//   2. ordered compaction: every thread owns a CONTIGUOUS index range, counts its elements above the
//      threshold T and equal to it, a block scan turns the counts into output slots, so the ties that
//      make it are the lowest-indexed ones;
//   3. bitonic sort of the (at most 1024) survivors in LDS by (value desc, index asc).
// Rows of up to 65536 scores are read from memory ONCE and stay in registers (16 or 64 per thread); longer rows are
// re-read in every pass (five reads).
#include "nw_internal.h"

namespace nw {
namespace {

constexpr int TK_THREADS = 1024;
constexpr int TK_MAXK = 1024;

__device__ __forceinline__ unsigned ordered_bits(float f) {  // larger float <=> larger uint; NaN on top
    unsigned b = __float_as_uint(f);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // every NaN, either sign: one image above +inf (torch's order)
    if ((b << 1) == 0) b = 0;  // -0.0 == +0.0: one image, so that their order is the index order
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

constexpr int TK_U = 8;       // independent loads in flight per thread in the histogram passes (long rows: 254 -> ... us at N = 50000 from 4)
// REGS > 0: the row is read from memory ONCE.  Wave w owns the contiguous segment [w seg, (w + 1) seg) of the row
// (seg a multiple of 64, 16 seg >= N) and keeps it in registers, element e of lane l = w seg + 64 e + l (coalesced
// loads); the histogram passes and the ordered compaction (ranks from ballots inside the wave, wave bases from one
// LDS round) all work from those registers.  REGS = 0 (N > 65536): the row is re-read in every pass.
template <int REGS>
__global__ __launch_bounds__(TK_THREADS) void nw_topk_kernel(const float* __restrict__ scores,
                                                             int64_t* __restrict__ idx_out,
                                                             float* __restrict__ val_out, int N, int k) {
    constexpr bool CACHED = REGS > 0;
    constexpr int TK_REGS = CACHED ? REGS : TK_U;
    __shared__ unsigned hist[256];
    __shared__ unsigned sel_prefix, sel_need;  // prefix of the k-th largest so far; how many of its bin are still needed
    __shared__ unsigned scan_gt[TK_THREADS / 64], scan_eq[TK_THREADS / 64];
    __shared__ unsigned cand_u[TK_MAXK];
    __shared__ int cand_i[TK_MAXK];
    const int tid = threadIdx.x, lane = tid & 63;
    const float* row = scores + (size_t)blockIdx.x * N;

    // ---- 1. radix select of the k-th largest
    if (tid == 0) {
        sel_prefix = 0;
        sel_need = (unsigned)k;
    }
    unsigned ureg[TK_REGS];   // CACHED: element wave * seg + 64 e + lane of the row
    const int wv = tid >> 6;
    const int seg = ((N + 16 * 64 - 1) / (16 * 64)) * 64;   // elements per wave
    if (CACHED) {
#pragma unroll
        for (int e = 0; e < TK_REGS; ++e) {   // coalesced, independent loads; the only pass over memory
            const int i = wv * seg + 64 * e + lane;
            ureg[e] = (64 * e < seg && i < N) ? ordered_bits(row[i]) : 0u;
        }
    }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = sel_prefix;
        // one element into the histogram: lanes that hit the same bin are merged first (two rounds take care of a hot bin),
        // then plain atomics; only the top byte (sign + high exponent bits) is that concentrated -- the lower digits are
        // spread over the bins and the merge would cost more than the atomics it saves
        auto count = [&](unsigned u, bool ok) {
            bool live = ok && ((pass == 0) || ((u >> (shift + 8)) == (prefix >> (shift + 8))));
            const unsigned bin = (u >> shift) & 255u;
            for (int round = 0; round < (pass == 0 ? 2 : 0); ++round) {
                const unsigned long long act = __ballot(live);
                if (!act) break;
                const unsigned b0 = __builtin_amdgcn_readlane(bin, __builtin_ctzll(act));
                const unsigned long long same = __ballot(live && bin == b0);
                if (live && bin == b0) {
                    if (lane == __builtin_ctzll(same)) atomicAdd(&hist[b0], (unsigned)__builtin_popcountll(same));
                    live = false;
                }
            }
            if (live) atomicAdd(&hist[bin], 1u);
        };
        if (CACHED) {
#pragma unroll
            for (int e = 0; e < TK_REGS; ++e) {
                count(ureg[e], ureg[e] != 0u);   // 0 is the image of no float (the smallest, -inf, is 0x007fffff): "not an element"
                __builtin_amdgcn_sched_barrier(0);   // (one element at a time: hoisted ballots spill the scalar file)
            }
        } else {
            for (int i0 = 0; i0 < N; i0 += TK_U * TK_THREADS) {
                unsigned u4[TK_U];
                bool ok[TK_U];
#pragma unroll
                for (int e = 0; e < TK_U; ++e) {  // coalesced, independent: TK_U loads in flight
                    const int i = i0 + e * TK_THREADS + tid;
                    ok[e] = i < N;
                    u4[e] = ok[e] ? ordered_bits(row[i]) : 0u;
                }
#pragma unroll
                for (int e = 0; e < TK_U; ++e) count(u4[e], ok[e]);
            }
        }
        __syncthreads();
        if (tid < 64) {  // one wave walks the 256 bins from the top: 4 bins per lane
            unsigned c[4], tot = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[j] = hist[255 - (4 * tid + j)];
                tot += c[j];
            }
            unsigned incl = tot;  // inclusive scan over lanes (lane 0 = top bins)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            unsigned before = incl - tot;  // elements in bins above this lane's four
            const unsigned need = sel_need;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (before < need && before + c[j] >= need) {  // exactly one (lane, j) satisfies this
                    sel_prefix = prefix | ((unsigned)(255 - (4 * tid + j)) << shift);
                    sel_need = need - before;
                }
                before += c[j];
            }
        }
        __syncthreads();
    }
    const unsigned T = sel_prefix;       // ordered bits of the k-th largest element
    const unsigned n_eq = sel_need;      // how many elements equal to T make it (the lowest-indexed ones)
    const unsigned n_gt = (unsigned)k - n_eq;

    // ---- 2. ordered compaction (ties: the lowest indices make it)
    if (CACHED) {
        // from the registers: every wave counts its segment, one LDS round gives the waves' bases, then the wave walks
        // its elements in index order and ranks the survivors of each 64 with ballots
        unsigned cg = 0, ce = 0;
#pragma unroll
        for (int e = 0; e < TK_REGS; ++e) {
            // (slots past the row hold 0, below every image and so below T: they drop out by themselves)
            cg += (unsigned)__builtin_popcountll(__ballot(ureg[e] > T));
            ce += (unsigned)__builtin_popcountll(__ballot(ureg[e] == T));
            __builtin_amdgcn_sched_barrier(0);   // (one ballot pair at a time: hoisted together they spill the scalar file)
        }
        if (lane == 0) {
            scan_gt[wv] = cg;
            scan_eq[wv] = ce;
        }
        __syncthreads();
        unsigned og = 0, oe = 0;
        for (int w = 0; w < wv; ++w) {
            og += scan_gt[w];
            oe += scan_eq[w];
        }
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int e = 0; e < TK_REGS; ++e) {
            const int i = wv * seg + 64 * e + lane;
            const bool isg = ureg[e] > T, ise = ureg[e] == T;
            const unsigned long long mg = __ballot(isg), me = __ballot(ise);
            if (isg) {
                const unsigned slot = og + (unsigned)__builtin_popcountll(mg & below);
                cand_u[slot] = ureg[e];
                cand_i[slot] = i;
            } else if (ise) {
                const unsigned slot = oe + (unsigned)__builtin_popcountll(me & below);
                if (slot < n_eq) {
                    cand_u[n_gt + slot] = ureg[e];
                    cand_i[n_gt + slot] = i;
                }
            }
            og += (unsigned)__builtin_popcountll(mg);
            oe += (unsigned)__builtin_popcountll(me);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
    // contiguous per-thread index ranges, the row read twice more
    const int per = (N + TK_THREADS - 1) / TK_THREADS;
    const int lo = min(tid * per, N), hi = min(lo + per, N);
    unsigned cg = 0, ce = 0;
    for (int i = lo; i < hi; ++i) {
        const unsigned u = ordered_bits(row[i]);
        cg += u > T;
        ce += u == T;
    }
    // exclusive block scans of the two counts: inside a wave by shuffles, across the 16 waves through LDS (two
    // barriers; a Hillis-Steele scan over the 1024 threads took twenty)
    unsigned ig = cg, ie = ce;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned a = __shfl_up(ig, o), b = __shfl_up(ie, o);
        if (lane >= o) {
            ig += a;
            ie += b;
        }
    }
    if (lane == 63) {
        scan_gt[wv] = ig;
        scan_eq[wv] = ie;
    }
    __syncthreads();
    unsigned og = ig - cg, oe = ie - ce;
    for (int w = 0; w < wv; ++w) {
        og += scan_gt[w];
        oe += scan_eq[w];
    }
    for (int i = lo; i < hi; ++i) {
        const unsigned u = ordered_bits(row[i]);
        if (u > T) {
            cand_u[og] = u;
            cand_i[og] = i;
            ++og;
        } else if (u == T) {
            if (oe < n_eq) {
                cand_u[n_gt + oe] = u;
                cand_i[n_gt + oe] = i;
            }
            ++oe;
        }
    }
    }
    __syncthreads();

    // ---- 3. bitonic sort of the k survivors: value descending, index ascending
    int P = 1;
    while (P < k) P <<= 1;
    for (int x = k + tid; x < P; x += TK_THREADS) {  // padding sorts last
        cand_u[x] = 0;
        cand_i[x] = 0x7fffffff;
    }
    __syncthreads();
    auto before = [](unsigned ua, int ia, unsigned ub, int ib) { return ua > ub || (ua == ub && ia < ib); };
    if (P <= 64) {
        // up to 64 survivors: one wave sorts them in registers (a bitonic network of shuffles, no barriers; the LDS
        // version below costs a block-wide barrier per stage, ten of them at k = 10)
        if (tid < 64) {
            unsigned u = tid < P ? cand_u[tid] : 0u;
            int ix = tid < P ? cand_i[tid] : 0x7fffffff;
            for (int size = 2; size <= 64; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const unsigned uo = __shfl_xor(u, stride);
                    const int io = __shfl_xor(ix, stride);
                    const bool lower = (tid & stride) == 0;            // this lane keeps the pair's first element ...
                    const bool up = (tid & size) == 0;                 // ... of a "best first" run
                    const bool mine_first = before(u, ix, uo, io);
                    const bool keep = (lower == up) ? mine_first : !mine_first;
                    if (!keep) {
                        u = uo;
                        ix = io;
                    }
                }
            if (tid < k) {
                idx_out[(size_t)blockIdx.x * k + tid] = ix;
                if (val_out) val_out[(size_t)blockIdx.x * k + tid] = row[ix];
            }
        }
        return;
    }
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int x = tid; x < P / 2; x += TK_THREADS) {
                const int a = 2 * x - (x & (stride - 1)), b = a + stride;
                const bool up = ((a & size) == 0);  // this pair's run is sorted "best first"
                const unsigned ua = cand_u[a], ub = cand_u[b];
                const int ia = cand_i[a], ib = cand_i[b];
                if (before(ub, ib, ua, ia) == up) {
                    cand_u[a] = ub; cand_i[a] = ib;
                    cand_u[b] = ua; cand_i[b] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (int x = tid; x < k; x += TK_THREADS) {
        idx_out[(size_t)blockIdx.x * k + x] = cand_i[x];
        if (val_out) val_out[(size_t)blockIdx.x * k + x] = row[cand_i[x]];
    }
}

}  // namespace

int launch_topk(const float* scores, int64_t* idx, float* vals, int64_t B, int64_t N, int64_t k, hipStream_t st) {
    if (k < 1 || k > N || k > TK_MAXK || N >= (1ll << 31) || B >= (1ll << 31)) return NW_ERR_UNSUPPORTED;
    if (B == 0) return NW_OK;
    if (N <= 16 * TK_THREADS)
        hipLaunchKernelGGL((nw_topk_kernel<16>), dim3((unsigned)B), dim3(TK_THREADS), 0, st, scores, idx, vals, (int)N, (int)k);
    else if (N <= 32 * TK_THREADS)
        hipLaunchKernelGGL((nw_topk_kernel<32>), dim3((unsigned)B), dim3(TK_THREADS), 0, st, scores, idx, vals, (int)N, (int)k);
    else if (N <= 64 * TK_THREADS)
        hipLaunchKernelGGL((nw_topk_kernel<64>), dim3((unsigned)B), dim3(TK_THREADS), 0, st, scores, idx, vals, (int)N, (int)k);
    else
        hipLaunchKernelGGL((nw_topk_kernel<0>), dim3((unsigned)B), dim3(TK_THREADS), 0, st, scores, idx, vals, (int)N, (int)k);
    NW_CHECK_LAUNCH();
    return NW_OK;
}

}  // namespace nw
