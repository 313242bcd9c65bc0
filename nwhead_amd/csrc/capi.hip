// capi.hip -- the extern "C" surface declared in include/nwhead_hip.h (gfx950 / MI355X only).
#include <cstddef>
#include <cstdlib>
#include <string.h>
#include "nw_internal.h"

namespace {
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline bool bad_kind(int kind) { return kind < NW_SCORE_EUCLIDEAN || kind > NW_SCORE_CLIP; }
}  // namespace

// Below ~2e8 multiply-adds the fp32-MFMA path with cached norms was the shorter one while the split-fp16 path
// had a query-split launch in front (measured then: B=64 N=1000 d=512 19.7 vs 29.8 us; B=256 N=10000 d=512
// 36.1 vs 23.9 us).  nw_fwd_opts.force_split takes the split path at every size.
static bool split_pays(int64_t B, int64_t N, int64_t d) {
    return nw::fwd_opts().force_split || (double)B * (double)N * (double)d >= 2.0e8;
}

namespace nw {
namespace {
thread_local FwdOpts tl_fwd_opts;     // of the forward entry point this thread is inside (OptsGuard), defaults outside
int g_knobs[KNOB_COUNT];
bool g_knobs_init = [] { for (int& k : g_knobs) k = KNOB_UNSET; return true; }();
const char* const knob_names[KNOB_COUNT] = {
    "pvar", "qg", "tile_rs", "merge_mq", "merge_per_query", "merge_no_global_tables", "persistent_any_rs", "no_persistent",
    "split_queries", "bwd_no_mfma", "bwd_split", "coeff_threads", "xgemm_wgs", "xgemm_nbuf", "split_lbits", "conv_gather",
    "conv_max_wgs", "wgrad_min_stages", "conv_skip_cfgs", "wgrad_batch_wgs", "bn_inline_fin", "conv_moments_per_tile",
    "conv_force_cfg"};
}  // namespace
const FwdOpts& fwd_opts() { return tl_fwd_opts; }
int knob(int id) { return (id >= 0 && id < KNOB_COUNT) ? __atomic_load_n(&g_knobs[id], __ATOMIC_RELAXED) : KNOB_UNSET; }
}  // namespace nw

namespace {
struct OptsGuard {   // the call's options are visible to the launch code of this thread until the entry point returns
    nw::FwdOpts saved;
    explicit OptsGuard(const nw_fwd_opts* o) : saved(nw::tl_fwd_opts) {
        nw::FwdOpts f;
        // (a caller built against an older header passes a shorter struct: the fields it has are honoured, and its tables,
        //  whose label array is then unknown, are not used)
        if (o && o->struct_size >= offsetof(nw_fwd_opts, tables)) {
            f.persistent_wgs = o->persistent_wgs;
            f.force_split = o->force_split;
        }
        if (o && o->struct_size >= sizeof(nw_fwd_opts)) {
            f.tables = static_cast<const char*>(o->tables);
            f.tables_bytes = o->tables_bytes;
            f.tables_sy = o->tables_sy;
            f.tables_N = o->tables_N;
        }
        nw::tl_fwd_opts = f;
    }
    ~OptsGuard() { nw::tl_fwd_opts = saved; }
};
}  // namespace

extern "C" int nw_debug_set(const char* name, int value) {
    if (!name) return NW_ERR_INVALID_ARG;
    for (int k = 0; k < nw::KNOB_COUNT; ++k)
        if (!strcmp(name, nw::knob_names[k])) {
            __atomic_store_n(&nw::g_knobs[k], value, __ATOMIC_RELAXED);
            return NW_OK;
        }
    return NW_ERR_INVALID_ARG;
}

extern "C" int nw_abi_version(void) { return NW_ABI_VERSION; }

extern "C" const char* nw_status_string(int status) {
    switch (status) {
        case NW_OK: return "ok";
        case NW_ERR_INVALID_ARG: return "invalid argument";
        case NW_ERR_UNSUPPORTED: return "unsupported score kind or size";
        case NW_ERR_WORKSPACE: return "workspace too small";
        case NW_ERR_LAUNCH: return "HIP launch failed";
        case NW_ERR_NO_DEVICE: return "no gfx950 device";
        default: return "unknown status";
    }
}

extern "C" int nw_device_check(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return NW_ERR_NO_DEVICE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NW_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NW_ERR_NO_DEVICE;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? NW_OK : NW_ERR_NO_DEVICE;
}

extern "C" int nw_scores_f32(const float* q, const float* s, float* scores, int64_t B, int64_t N,
                             int64_t d, int kind, const float* logit_scale_dev, int sup_batched,
                             void* stream) {
    if (B < 0 || N < 0 || d < 0) return NW_ERR_INVALID_ARG;
    if (bad_kind(kind)) return NW_ERR_UNSUPPORTED;
    if (B == 0 || N == 0) return NW_OK;
    if (!q || !s || !scores) return NW_ERR_INVALID_ARG;
    if (kind == NW_SCORE_CLIP && !logit_scale_dev) return NW_ERR_INVALID_ARG;
    return nw::launch_scores(q, s, scores, B, N, d, kind, logit_scale_dev, sup_batched,
                             static_cast<hipStream_t>(stream));
}

// Scratch: the fused path keeps per-tile softmax statistics and label-run sums (fused.hip); the
// two-kernel path (per-query supports, N <= 25, weights requested) one (B,N) score matrix.
static size_t agg_slice_bytes(int64_t B, int64_t N, int64_t C) {
    const int S = nw::aggregate_slices(B, N);
    return S > 1 ? align256((size_t)S * (size_t)B * (size_t)(2 + (C > 0 ? C : 0)) * sizeof(float)) : 0;
}

extern "C" size_t nw_fwd_workspace_bytes(int64_t B, int64_t N, int64_t d, int64_t C) {
    (void)d; (void)C;
    if (B <= 0 || N <= 0) return 0;
    const size_t plain = align256((size_t)B * (size_t)N * sizeof(float));
    const size_t fused = align256(nw::fused_workspace_bytes(B, N, d, C));
    // + room for the split-fp16 form of the queries (rows, scales, norms) behind the fused area
    const size_t qsplit = align256((size_t)B * (size_t)d * sizeof(float)) + 2 * align256((size_t)B * sizeof(float));
    // + B floats for the log-sum-exp when weights / influences are derived from the fused kernel's scores
    // + the slice partials of the two-kernel path's aggregation (few queries, long rows: nw::aggregate_slices)
    return (plain > fused ? plain : fused) + qsplit + align256((size_t)B * sizeof(float)) + agg_slice_bytes(B, N, C);
}

extern "C" int nw_row_norm2_f32(const float* x, float* n2, int64_t rows, int64_t d, void* stream) {
    if (rows < 0 || d < 0) return NW_ERR_INVALID_ARG;
    if (rows == 0) return NW_OK;
    if (!x || !n2) return NW_ERR_INVALID_ARG;
    return nw::launch_rownorm2(x, n2, rows, d, static_cast<hipStream_t>(stream));
}

extern "C" int nw_fwd_f32(const float* q, const float* s, const int64_t* sy, const float* s_norm2,
                          const float* s_split, const float* s_scale, float* out, float* scores_out, float* lse_out, float* weights_out, void* workspace,
                          size_t workspace_bytes, int64_t B, int64_t N, int64_t d, int64_t C,
                          int kind, const float* logit_scale_dev, int sup_batched,
                          int labels_batched, const nw_fwd_opts* opts, void* stream) {
    OptsGuard opts_guard(opts);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B < 0 || N < 0 || d < 0 || C < 0) return NW_ERR_INVALID_ARG;
    if (bad_kind(kind)) return NW_ERR_UNSUPPORTED;
    if (B == 0) return NW_OK;
    if (!out && C > 0) return NW_ERR_INVALID_ARG;
    if (N > 0 && (!q || !s || !sy)) return NW_ERR_INVALID_ARG;
    if (kind == NW_SCORE_CLIP && !logit_scale_dev) return NW_ERR_INVALID_ARG;
    if (labels_batched && !sup_batched) return NW_ERR_INVALID_ARG;
    // Softmax weights on request: the fused kernel writes the scores where the weights go and one in-place pass
    // normalises them with the merge's log-sum-exp (w = exp(score - lse)); the two-kernel fallback below streams
    // the (B,N) matrix five times.
    float* sc_buf = scores_out ? scores_out : weights_out;
    if (N > 0 && C > 0 && !sup_batched && nw::fused_eligible(q, s, B, N, d, C) &&
        (!sc_buf || (reinterpret_cast<uintptr_t>(sc_buf) & 15) == 0)) {
        if (!workspace || workspace_bytes < nw_fwd_workspace_bytes(B, N, d, C)) return NW_ERR_WORKSPACE;
        if (weights_out) {
            float* lse = lse_out;
            if (!lse) lse = reinterpret_cast<float*>(static_cast<char*>(workspace) + nw_fwd_workspace_bytes(B, N, d, C) -
                                                     align256((size_t)B * sizeof(float)) - agg_slice_bytes(B, N, C));
            const int rc = nw_fwd_f32(q, s, sy, s_norm2, s_split, s_scale, out, sc_buf, lse, nullptr, workspace,
                                      workspace_bytes, B, N, d, C, kind, logit_scale_dev, 0, 0, opts, stream);
            if (rc != NW_OK) return rc;
            return nw::launch_weights_from_scores(sc_buf, lse, weights_out, B, N, st);
        }
        if (s_split && s_scale && s_norm2 && d % 32 == 0 && split_pays(B, N, d) &&
            ((reinterpret_cast<uintptr_t>(s_split) | reinterpret_cast<uintptr_t>(q)) & 15) == 0) {
            return nw::launch_fused(q, s_split, sy, s_norm2, s_scale, logit_scale_dev,
                                    out, scores_out, lse_out, nullptr, nullptr, nullptr, workspace,
                                    workspace_bytes, B, N, d, C, kind, st);
        }
        return nw::launch_fused(q, s, sy, s_norm2, nullptr, logit_scale_dev, out, scores_out,
                                lse_out, nullptr, nullptr, nullptr, workspace, workspace_bytes, B, N, d, C, kind, st);
    }
    float* scores = scores_out;
    if (!scores && N > 0) {
        if (!workspace || workspace_bytes < nw_fwd_workspace_bytes(B, N, d, C)) return NW_ERR_WORKSPACE;
        scores = static_cast<float*>(workspace);
    }
    int rc = nw::launch_scores(q, s, scores, B, N, d, kind, logit_scale_dev, sup_batched, st);
    if (rc != NW_OK) return rc;
    float* slice_ws = nullptr;
    size_t slice_floats = 0;
    const size_t sb = agg_slice_bytes(B, N, C);
    if (sb && workspace && workspace_bytes >= nw_fwd_workspace_bytes(B, N, d, C)) {   // (the last area of the workspace)
        slice_ws = reinterpret_cast<float*>(static_cast<char*>(workspace) + nw_fwd_workspace_bytes(B, N, d, C) - sb);
        slice_floats = sb / sizeof(float);
    }
    return nw::launch_aggregate(scores, sy, labels_batched, out, lse_out, weights_out, nullptr,
                                nullptr, nullptr, B, N, C, st, slice_ws, slice_floats);
}

extern "C" int nw_fwd_partial_f32(const float* q, const float* s, const int64_t* sy,
                                  const float* s_norm2, const float* s_split, const float* s_scale,
                                  float* m, float* den, float* num, void* workspace, size_t workspace_bytes,
                                  int64_t B, int64_t N, int64_t d, int64_t C, int kind,
                                  const float* logit_scale_dev, const nw_fwd_opts* opts, void* stream) {
    OptsGuard opts_guard(opts);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B < 0 || N < 0 || d < 0 || C < 0) return NW_ERR_INVALID_ARG;
    if (bad_kind(kind)) return NW_ERR_UNSUPPORTED;
    if (B == 0) return NW_OK;
    if (!m || !den || (!num && C > 0)) return NW_ERR_INVALID_ARG;
    if (N > 0 && (!q || !s || !sy)) return NW_ERR_INVALID_ARG;
    if (kind == NW_SCORE_CLIP && !logit_scale_dev) return NW_ERR_INVALID_ARG;
    float* scores = static_cast<float*>(workspace);
    if (N > 0 && (!workspace || workspace_bytes < nw_fwd_workspace_bytes(B, N, d, C))) return NW_ERR_WORKSPACE;
    if (N > 0 && C > 0 && nw::fused_eligible(q, s, B, N, d, C)) {
        if (s_split && s_scale && s_norm2 && d % 32 == 0 && split_pays(B, N, d) &&
            ((reinterpret_cast<uintptr_t>(s_split) | reinterpret_cast<uintptr_t>(q)) & 15) == 0) {
            return nw::launch_fused(q, s_split, sy, s_norm2, s_scale, logit_scale_dev,
                                    nullptr, nullptr, nullptr, m, den, num, workspace, workspace_bytes, B, N, d,
                                    C, kind, st);
        }
        return nw::launch_fused(q, s, sy, s_norm2, nullptr, logit_scale_dev, nullptr, nullptr,
                                nullptr, m, den, num, workspace, workspace_bytes, B, N, d, C, kind, st);
    }
    int rc = nw::launch_scores(q, s, scores, B, N, d, kind, logit_scale_dev, 0, st);
    if (rc != NW_OK) return rc;
    return nw::launch_aggregate(scores, sy, 0, nullptr, nullptr, nullptr, m, den, num, B, N, C, st);
}

extern "C" int nw_merge_finalize_f32(const float* m, const float* den, const float* num, float* out,
                                     int64_t G, int64_t B, int64_t C, int64_t stride_m,
                                     int64_t stride_den, int64_t stride_num,
                                     const int64_t* class_lo, int64_t C_local, void* stream) {
    if (G < 0 || B < 0 || C < 0) return NW_ERR_INVALID_ARG;
    if (B == 0 || C == 0) return NW_OK;
    if (!m || !den || !num || !out) return NW_ERR_INVALID_ARG;
    if (class_lo && C_local <= 0) return NW_ERR_INVALID_ARG;
    return nw::launch_merge(m, den, num, out, G, B, C, stride_m, stride_den, stride_num, class_lo,
                            class_lo ? C_local : C, static_cast<hipStream_t>(stream));
}

extern "C" int nw_topk_f32(const float* scores, int64_t* idx_out, float* val_out, int64_t B, int64_t N, int64_t k,
                           void* stream) {
    if (B < 0 || N < 0 || (B > 0 && (!scores || !idx_out))) return NW_ERR_INVALID_ARG;
    return nw::launch_topk(scores, idx_out, val_out, B, N, k, static_cast<hipStream_t>(stream));
}

extern "C" int nw_debug_tile_timing(int enable) { return nw::tile_timer_enable(enable != 0); }

extern "C" int nw_debug_tile_timing_read(double* total_us, int64_t* launches) {
    return nw::tile_timer_read(total_us, launches);
}

extern "C" int nw_fwd_influence_f32(const float* q, const float* s, const int64_t* sy, const float* s_norm2,
                                    const float* s_split, const float* s_scale, const int64_t* qy, float* out,
                                    float* lse_out, float* infl_out, void* workspace, size_t workspace_bytes,
                                    int64_t B, int64_t N, int64_t d, int64_t C, int kind,
                                    const float* logit_scale_dev, const nw_fwd_opts* opts, void* stream) {
    OptsGuard opts_guard(opts);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B < 0 || N < 0 || d < 0 || C < 0) return NW_ERR_INVALID_ARG;
    if (B == 0) return NW_OK;
    if (N == 0 || C == 0)   // no supports: log(0 + 1e-12) and nothing to score
        return nw_fwd_f32(q, s, sy, s_norm2, s_split, s_scale, out, nullptr, lse_out, nullptr, workspace, workspace_bytes,
                          B, N, d, C, kind, logit_scale_dev, 0, 0, opts, stream);
    if (!qy || !infl_out || !out) return NW_ERR_INVALID_ARG;
    const size_t need = nw_fwd_workspace_bytes(B, N, d, C);   // its last B floats: the log-sum-exp slot
    if (!workspace || workspace_bytes < need) return NW_ERR_WORKSPACE;
    float* lse = lse_out ? lse_out : reinterpret_cast<float*>(static_cast<char*>(workspace) + need - align256((size_t)B * sizeof(float)) -
                                                              agg_slice_bytes(B, N, C));
    // scores into infl_out (fused tile kernel when eligible, else the score kernel of the two-kernel path)
    const int rc = nw_fwd_f32(q, s, sy, s_norm2, s_split, s_scale, out, infl_out, lse, nullptr, workspace,
                              workspace_bytes, B, N, d, C, kind, logit_scale_dev, 0, 0, opts, stream);
    if (rc != NW_OK) return rc;
    return nw::launch_influence(out, qy, infl_out, sy, lse, infl_out, B, N, C, st);
}

extern "C" int nw_aggregate_f32(const float* scores, const int64_t* sy, float* out, float* lse_out, float* weights_out,
                                int64_t B, int64_t N, int64_t C, int labels_batched, void* stream) {
    if (B < 0 || N < 0 || C < 0) return NW_ERR_INVALID_ARG;
    if (B == 0) return NW_OK;
    if ((!out && C > 0) || (N > 0 && (!scores || !sy))) return NW_ERR_INVALID_ARG;
    return nw::launch_aggregate(scores, sy, labels_batched, out, lse_out, weights_out, nullptr, nullptr, nullptr, B, N, C,
                                static_cast<hipStream_t>(stream));
}
