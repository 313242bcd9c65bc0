// Host-only exercise of the C ABI's argument validation (include/nwhead_hip.h) under AddressSanitizer +
// UndefinedBehaviorSanitizer.  Built by `make -C nwhead_amd/csrc sanitize` from the same sources
// as libnwhead_hip.so with --cuda-host-only: no device code, no GPU needed -- every call here must be refused (or be a
// pure host computation) BEFORE anything is launched, with the documented status and without a sanitizer report.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/nwhead_hip.h"

static int failures = 0;
#define EXPECT(call, want)                                                                  \
    do {                                                                                    \
        const long long got_ = (long long)(call);                                          \
        if (got_ != (long long)(want)) {                                                    \
            std::printf("FAIL %s:%d  %s = %lld, expected %s\n", __FILE__, __LINE__, #call, got_, #want); \
            ++failures;                                                                     \
        }                                                                                   \
    } while (0)

int main() {
    std::vector<float> f(4096, 1.f);
    std::vector<int64_t> y(64, 0);
    float* F = f.data();
    int64_t* Y = y.data();
    alignas(16) char ws[256];
    EXPECT(nw_abi_version(), NW_ABI_VERSION);
    EXPECT(std::strcmp(nw_status_string(NW_OK), "ok"), 0);
    EXPECT(nw_status_string(12345) != nullptr, 1);
    // sizes: pure host arithmetic, monotone, no overflow for large shapes
    EXPECT(nw_fwd_workspace_bytes(0, 10, 4, 3), 0);
    EXPECT(nw_fwd_workspace_bytes(8, 0, 4, 3), 0);
    EXPECT(nw_fwd_workspace_bytes(8, 64, 128, 10) > 0, 1);
    EXPECT(nw_fwd_workspace_bytes(4096, 50000, 512, 200) >= (size_t)4096 * 50000 * 4, 1);
    EXPECT(nw_fwd_workspace_bytes(1 << 20, 1 << 20, 512, 200) > ((size_t)1 << 40), 1);
    EXPECT(nw_bwd_workspace_bytes(0, 5, 4, 3, 0, 0), 0);
    EXPECT(nw_bwd_workspace_bytes(256, 10000, 512, 200, 0, 0) > 0, 1);
    // scores
    EXPECT(nw_scores_f32(F, F, F, -1, 4, 4, NW_SCORE_EUCLIDEAN, nullptr, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_scores_f32(F, F, F, 4, 4, 4, 99, nullptr, 0, nullptr), NW_ERR_UNSUPPORTED);
    EXPECT(nw_scores_f32(F, F, F, 0, 4, 4, NW_SCORE_EUCLIDEAN, nullptr, 0, nullptr), NW_OK);
    EXPECT(nw_scores_f32(nullptr, F, F, 4, 4, 4, NW_SCORE_EUCLIDEAN, nullptr, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_scores_f32(F, F, F, 4, 4, 4, NW_SCORE_CLIP, nullptr, 0, nullptr), NW_ERR_INVALID_ARG);
    // forward
    EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, ws, sizeof ws, -1, 4, 4, 3,
                      NW_SCORE_EUCLIDEAN, nullptr, 0, 0, nullptr, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, ws, sizeof ws, 4, 4, 4, 3, -3,
                      nullptr, 0, 0, nullptr, nullptr), NW_ERR_UNSUPPORTED);
    EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, sizeof ws, 4, 4, 4, 3,
                      NW_SCORE_EUCLIDEAN, nullptr, 0, 0, nullptr, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_fwd_f32(nullptr, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, ws, sizeof ws, 4, 4, 4, 3,
                      NW_SCORE_EUCLIDEAN, nullptr, 0, 0, nullptr, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, ws, sizeof ws, 4, 4, 4, 3,
                      NW_SCORE_CLIP, nullptr, 0, 0, nullptr, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, ws, sizeof ws, 4, 4, 4, 3,
                      NW_SCORE_EUCLIDEAN, nullptr, 0, 1, nullptr, nullptr), NW_ERR_INVALID_ARG);   // batched labels, shared support
    EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, nullptr, 0, 0, 4, 4, 3,
                      NW_SCORE_EUCLIDEAN, nullptr, 0, 0, nullptr, nullptr), NW_OK);                // B == 0
    EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, ws, sizeof ws, 8, 64, 16, 3,
                      NW_SCORE_EUCLIDEAN, nullptr, 0, 0, nullptr, nullptr), NW_ERR_WORKSPACE);     // fused path, workspace too small
    EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, nullptr, 0, 8, 64, 16, 3,
                      NW_SCORE_EUCLIDEAN, nullptr, 0, 0, nullptr, nullptr), NW_ERR_WORKSPACE);
    // partials / merge
    EXPECT(nw_fwd_partial_f32(F, F, Y, nullptr, nullptr, nullptr, nullptr, F, F, ws, sizeof ws, 4, 4, 4, 3,
                              NW_SCORE_EUCLIDEAN, nullptr, nullptr, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_fwd_partial_f32(F, F, Y, nullptr, nullptr, nullptr, F, F, F, ws, 8, 8, 64, 16, 3, NW_SCORE_EUCLIDEAN,
                              nullptr, nullptr, nullptr), NW_ERR_WORKSPACE);
    EXPECT(nw_merge_finalize_f32(F, F, F, F, -1, 4, 3, 4, 4, 12, nullptr, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_merge_finalize_f32(F, F, F, nullptr, 2, 4, 3, 4, 4, 12, nullptr, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_merge_finalize_f32(F, F, F, F, 2, 4, 3, 4, 4, 12, Y, 0, nullptr), NW_ERR_INVALID_ARG);  // window without width
    EXPECT(nw_merge_finalize_f32(F, F, F, F, 2, 0, 3, 4, 4, 12, nullptr, 0, nullptr), NW_OK);
    // backward
    EXPECT(nw_bwd_f32(F, F, Y, F, F, F, F, F, F, nullptr, ws, sizeof ws, 4, 4, 4, 3, 77, nullptr, 0, 0, nullptr), NW_ERR_UNSUPPORTED);
    EXPECT(nw_bwd_f32(F, F, Y, F, F, F, F, nullptr, F, nullptr, ws, sizeof ws, 4, 4, 4, 3, NW_SCORE_EUCLIDEAN, nullptr, 0, 0,
                      nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bwd_f32(F, F, Y, F, F, F, F, F, F, nullptr, ws, 8, 4, 4, 4, 3, NW_SCORE_EUCLIDEAN, nullptr, 0, 0, nullptr),
           NW_ERR_WORKSPACE);
    EXPECT(nw_bwd_f32(F, F, Y, F, F, F, F, F, F, nullptr, ws, sizeof ws, 4, 4, 4, 3, NW_SCORE_CLIP, nullptr, 0, 0, nullptr),
           NW_ERR_INVALID_ARG);
    // ... with the supports' bank: all three pieces or none, shared supports and d % 32 == 0 only
    EXPECT(nw_bwd_bank_f32(F, F, F, F, nullptr, Y, F, F, F, F, F, F, nullptr, ws, sizeof ws, 4, 4, 32, 3, NW_SCORE_EUCLIDEAN,
                           nullptr, 0, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bwd_bank_f32(F, F, nullptr, F, F, Y, F, F, F, F, F, F, nullptr, ws, sizeof ws, 4, 4, 32, 3, NW_SCORE_EUCLIDEAN,
                           nullptr, 0, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bwd_bank_f32(F, F, F, F, F, Y, F, F, F, F, F, F, nullptr, ws, sizeof ws, 4, 4, 48, 3, NW_SCORE_EUCLIDEAN,
                           nullptr, 0, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bwd_bank_f32(F, F, F, F, F, Y, F, F, F, F, F, F, nullptr, ws, sizeof ws, 4, 4, 32, 3, NW_SCORE_EUCLIDEAN,
                           nullptr, 1, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bwd_uses_split(256, 10000, 512, 200, 0), 1);
    EXPECT(nw_bwd_uses_split(256, 10000, 512, 200, 1), 0);     // per-query supports
    EXPECT(nw_bwd_uses_split(256, 10000, 500, 200, 0), 0);     // d % 32 != 0
    EXPECT(nw_bwd_uses_split(32, 10, 1024, 10, 0), 0);         // a training episode: small
    EXPECT(nw_bwd_uses_split(16, 256, 1024, 10, 0), 1);
    EXPECT(nw_bwd_uses_split(256, 60000, 512, 200, 0), 0);     // a row of coefficients does not fit in LDS
    EXPECT(nw_bwd_uses_split(-1, 10, 32, 10, 0), 0);
    // NHWC bias + residual + ReLU
    EXPECT(nw_bias_act_nhwc_f32(F, F, nullptr, 1, F, 4, 6, nullptr), NW_ERR_UNSUPPORTED);       // c % 4
    EXPECT(nw_bias_act_nhwc_f32(nullptr, F, nullptr, 1, F, 4, 8, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bias_act_nhwc_f32(F, F, nullptr, 1, F, -1, 8, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bias_act_nhwc_f32(F, F, nullptr, 1, F, 0, 8, nullptr), NW_OK);
    // run tables of a resident bank
    EXPECT(nw_bank_tables_bytes(0), 0);
    EXPECT(nw_bank_tables_bytes(50000) >= (size_t)2 * 50000 * 4, 1);
    EXPECT(nw_bank_tables_build(nullptr, 10, 5, ws, sizeof ws, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bank_tables_build(Y, 10, 5, ws, 8, nullptr), NW_ERR_WORKSPACE);
    EXPECT(nw_bank_tables_build(Y, 0, 5, nullptr, 0, nullptr), NW_OK);
    {   // the options of a forward call: a short struct (older header) is ignored, tables too small are ignored
        nw_fwd_opts o = {};
        o.struct_size = 4; o.tables = ws; o.tables_bytes = 8;
        EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, nullptr, 0, 0, 4, 4, 3,
                          NW_SCORE_EUCLIDEAN, nullptr, 0, 0, &o, nullptr), NW_OK);
        o.struct_size = sizeof o;
        EXPECT(nw_fwd_f32(F, F, Y, nullptr, nullptr, nullptr, F, nullptr, nullptr, nullptr, ws, sizeof ws, 8, 64, 16, 3,
                          NW_SCORE_EUCLIDEAN, nullptr, 0, 0, &o, nullptr), NW_ERR_WORKSPACE);
    }
    EXPECT(nw_debug_set(nullptr, 1), NW_ERR_INVALID_ARG);
    EXPECT(nw_debug_set("no_such_knob", 1), NW_ERR_INVALID_ARG);
    EXPECT(nw_debug_set("qg", 8), NW_OK);
    // split rows, norms, influence, top-k, aggregate
    EXPECT(nw_split_rows_f16x2(F, F, F, F, 4, 48, nullptr), NW_ERR_UNSUPPORTED);      // d % 32 != 0
    EXPECT(nw_split_rows_f16x2(F, F, F, F, -1, 32, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_split_rows_f16x2(nullptr, F, F, F, 4, 32, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_split_rows_f16x2(F + 1, F, F, F, 4, 32, nullptr), NW_ERR_INVALID_ARG);  // misaligned
    EXPECT(nw_split_rows_f16x2(F, F, F, F, 0, 32, nullptr), NW_OK);
    EXPECT(nw_row_norm2_f32(F, F, -2, 4, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_row_norm2_f32(nullptr, F, 2, 4, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_support_influence_f32(F, Y, F, Y, F, -1, 4, 3, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_support_influence_f32(F, Y, nullptr, Y, F, 2, 4, 3, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_support_influence_f32(F, Y, F, Y, F, 0, 4, 3, nullptr), NW_OK);
    EXPECT(nw_fwd_influence_f32(F, F, Y, nullptr, nullptr, nullptr, nullptr, F, nullptr, F, ws, sizeof ws, 4, 8, 4, 3,
                                NW_SCORE_EUCLIDEAN, nullptr, nullptr, nullptr), NW_ERR_INVALID_ARG);   // no query labels
    EXPECT(nw_fwd_influence_f32(F, F, Y, nullptr, nullptr, nullptr, Y, F, nullptr, F, ws, 8, 4, 8, 4, 3,
                                NW_SCORE_EUCLIDEAN, nullptr, nullptr, nullptr), NW_ERR_WORKSPACE);
    EXPECT(nw_topk_f32(nullptr, Y, nullptr, 2, 8, 2, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_topk_f32(F, Y, nullptr, -1, 8, 2, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_aggregate_f32(nullptr, Y, F, nullptr, nullptr, 2, 8, 3, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_aggregate_f32(F, Y, F, nullptr, nullptr, -2, 8, 3, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_aggregate_bwd_f32(F, Y, F, F, F, nullptr, 2, 8, 3, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_aggregate_bwd_f32(F, Y, F, F, F, F, 2, 0, 3, 0, nullptr), NW_OK);
    // fused 1x1 convolution
    EXPECT(nw_conv1x1_f32(F, 64, nullptr, nullptr, 0, F, nullptr, 0, F, 64, nullptr, 0, -1, 4, 4, 16, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_conv1x1_f32(nullptr, 64, nullptr, nullptr, 0, F, nullptr, 0, F, 64, nullptr, 0, 1, 4, 4, 16, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_conv1x1_f32(F, 64, F, nullptr, 1, F, nullptr, 0, F, 64, nullptr, 0, 1, 4, 4, 16, nullptr), NW_ERR_INVALID_ARG);  // scale without shift
    EXPECT(nw_conv1x1_f32(F, 64, nullptr, nullptr, 0, F, nullptr, 0, F, 64, nullptr, 0, 1, 4, 6, 16, nullptr), NW_ERR_UNSUPPORTED);  // cout % 4
    EXPECT(nw_conv1x1_f32(F, 8, nullptr, nullptr, 0, F, nullptr, 0, F, 64, nullptr, 0, 1, 4, 4, 16, nullptr), NW_ERR_INVALID_ARG);   // batch stride < c*hw
    EXPECT(nw_conv1x1_f32(F, 64, nullptr, nullptr, 0, F, nullptr, 0, F, 64, nullptr, 0, 0, 4, 4, 16, nullptr), NW_OK);
    EXPECT(nw_conv1x1_workspace_bytes(64, 992, 128, 49) > 0, 1);               // 7x7 planes: 49 column tiles, K split
    EXPECT(nw_conv1x1_workspace_bytes(64, 992, 100, 49), 0);                   // cout % 128 != 0: the generic kernel, no K split
    EXPECT(nw_scale_shift_relu_avgpool2_f32(F, F, F, F, 1, 4, 4, 4, 8, 1, nullptr), NW_ERR_INVALID_ARG);   // batch stride < c*h*w
    EXPECT(nw_scale_shift_relu_avgpool2_f32(nullptr, F, F, F, 1, 4, 4, 4, 64, 1, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_scale_shift_relu_avgpool2_f32(F, F, F, F, 1, 4, 1, 4, 64, 1, nullptr), NW_OK);               // h / 2 == 0
    EXPECT(nw_conv1x1_workspace_bytes(64, 992, 128, 196) > 0, 1);              // 14x14: K split, partial tiles
    EXPECT(nw_conv1x1_workspace_bytes(64, 64, 128, 3136), 0);
    EXPECT(nw_conv1x1_f32(F, 992 * 196, nullptr, nullptr, 0, F, nullptr, 0, F, 128 * 196, nullptr, 0, 2, 992, 128, 196, nullptr),
           NW_ERR_WORKSPACE);
    // 3x3 convolution
    EXPECT(nw_conv3x3_f32(F, 64, F, nullptr, nullptr, 0, 0, F, 64, nullptr, 0, 1, 4, 32, -1, 4, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_conv3x3_f32(nullptr, 64, F, nullptr, nullptr, 0, 0, F, 512, nullptr, 0, 1, 4, 32, 4, 4, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_conv3x3_f32(F, 64, F, nullptr, nullptr, 0, 0, F, 512, nullptr, 0, 1, 4, 24, 4, 4, nullptr), NW_ERR_UNSUPPORTED);   // cout % 32
    EXPECT(nw_conv3x3_f32(F, 8, F, nullptr, nullptr, 0, 0, F, 512, nullptr, 0, 1, 4, 32, 4, 4, nullptr), NW_ERR_INVALID_ARG);     // batch stride < cin*H*W
    EXPECT(nw_conv3x3_f32(F, 64, F, nullptr, F, 8, 0, F, 512, nullptr, 0, 1, 4, 32, 4, 4, nullptr), NW_ERR_INVALID_ARG);         // residual stride
    EXPECT(nw_conv3x3_f32(F, 64, F, nullptr, nullptr, 0, 0, F, 512, nullptr, 0, 0, 4, 32, 4, 4, nullptr), NW_OK);
    EXPECT(nw_conv3x3_workspace_bytes(64, 512, 32, 7, 7) > 0, 1);                 // K split over workgroups
    EXPECT(nw_conv3x3_workspace_bytes(64, 128, 32, 56, 56), 0);
    EXPECT(nw_conv3x3_f32(F, 25088, F, nullptr, nullptr, 0, 0, F, 1568, nullptr, 0, 64, 512, 32, 7, 7, nullptr), NW_ERR_WORKSPACE);
    EXPECT(nw_conv3x3_workgroups(64, 512, 32, 7, 7) >= 192, 1);
    EXPECT(nw_conv3x3_workgroups(1, 8, 32, 7, 7), 1);
    // channels-last convolutions on the fp16 matrix cores, their weight gradient, BatchNorm in phases
    EXPECT(nw_conv2d_nhwc_supported(2, 14, 14, 128, 32, 3, 3, 1, 1), 1);
    EXPECT(nw_conv2d_nhwc_supported(2, 14, 14, 128, 24, 3, 3, 1, 1), 0);                       // Cout % 32
    EXPECT(nw_conv2d_nhwc_supported(2, 14, 14, 48, 32, 3, 3, 1, 1), 0);                        // Cin % 32 with KW * Cin > 32
    EXPECT(nw_conv2d_nhwc_f16x2(F, F, F, F, nullptr, nullptr, 0, F, nullptr, -1, 14, 14, 128, 32, 3, 3, 1, 1, 0, 0, nullptr, nullptr),
           NW_ERR_INVALID_ARG);
    EXPECT(nw_conv2d_nhwc_f16x2(F, F, F, F, nullptr, nullptr, 0, F, nullptr, 0, 14, 14, 128, 32, 3, 3, 1, 1, 0, 0, nullptr, nullptr), NW_OK);
    EXPECT(nw_conv2d_nhwc_f16x2(F, F, F, F, nullptr, nullptr, 0, F, nullptr, 2, 14, 14, 128, 24, 3, 3, 1, 1, 0, 0, nullptr, nullptr),
           NW_ERR_UNSUPPORTED);
    EXPECT(nw_conv2d_nhwc_f16x2(nullptr, F, F, F, nullptr, nullptr, 0, F, nullptr, 2, 14, 14, 128, 32, 3, 3, 1, 1, 0, 0, nullptr, nullptr),
           NW_ERR_INVALID_ARG);
    EXPECT(nw_conv2d_nhwc_f16x2(F, F, F, F, nullptr, nullptr, 0, F, nullptr, 2, 14, 14, 128, 32, 3, 3, 1, 1, 64, 0, nullptr, nullptr),
           NW_ERR_INVALID_ARG);                                                                 // ldx < Cin
    EXPECT(nw_conv2d_nhwc_f16x2(F, F, F, F, nullptr, nullptr, 0, F, nullptr, 2, 14, 14, 128, 32, 3, 3, 1, 1, 0, 34, nullptr, nullptr),
           NW_ERR_INVALID_ARG);                                                                 // ldy % 4
    EXPECT(nw_conv2d_nhwc_f16x2(F, F, F, F, nullptr, nullptr, 0, F, nullptr, 2, 14, 14, 4, 32, 7, 7, 2, 3, 8, 0, nullptr, nullptr),
           NW_ERR_UNSUPPORTED);                                                                 // a window of a few-channel input
    EXPECT(nw_conv2d_nhwc_f16x2(F, F, F, F, nullptr, nullptr, 0, F, nullptr, 2, 14, 14, 4, 32, 7, 7, 2, 3, 0, 0, F, nullptr),
           NW_ERR_UNSUPPORTED);                                                                 // moments of a few-channel input
    EXPECT(nw_conv2d_nhwc_moments_groups(42, 14, 14, 512, 128, 1, 1, 1, 0) > 0, 1);
    EXPECT(nw_conv2d_nhwc_moments_groups(42, 14, 14, 512, 100, 1, 1, 1, 0), 0);
    EXPECT(nw_conv2d_nhwc_moments_groups(0, 14, 14, 512, 128, 1, 1, 1, 0), 0);
    EXPECT(nw_conv2d_nhwc_wgrad_supported(2, 14, 14, 128, 32, 3, 3, 1, 1), 1);
    EXPECT(nw_conv2d_nhwc_wgrad_supported(2, 14, 14, 128, 32, 3, 3, 2, 1), 0);                 // strided
    EXPECT(nw_conv2d_nhwc_wgrad_supported(2, 64, 64, 128, 32, 3, 3, 1, 1), 0);                 // W > 62 with taps
    EXPECT(nw_conv2d_nhwc_wgrad_workspace_bytes(2, 14, 14, 128, 32, 3, 3, 1, 1) > 0, 1);
    EXPECT(nw_conv2d_nhwc_wgrad_f16x2(F, F, F, F, F, ws, 8, 2, 14, 14, 128, 32, 3, 3, 1, 1, 0, 0, nullptr), NW_ERR_WORKSPACE);
    EXPECT(nw_conv2d_nhwc_wgrad_f16x2(F, F, F, F, F, ws, sizeof ws, 2, 14, 14, 128, 32, 3, 3, 1, 1, 64, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_conv2d_nhwc_wgrad_f16x2(F, F, F, F, F, ws, sizeof ws, 2, 14, 14, 128, 32, 3, 3, 2, 1, 0, 0, nullptr), NW_ERR_UNSUPPORTED);
    EXPECT(nw_bn_nhwc_workspace_bytes(0, 64), 0);
    EXPECT(nw_bn_nhwc_workspace_bytes(8232, 128) > 0, 1);
    EXPECT(nw_bn_nhwc_moments_f32(F, 64, 100, 64, 1e-5f, F, F, F, ws, 8, nullptr), NW_ERR_WORKSPACE);
    EXPECT(nw_bn_nhwc_moments_f32(F, 32, 100, 64, 1e-5f, F, F, F, ws, sizeof ws, nullptr), NW_ERR_INVALID_ARG);   // ldx < c
    EXPECT(nw_bn_nhwc_moments_f32(F, 64, 100, 64, 1e-5f, nullptr, F, F, ws, sizeof ws, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bn_nhwc_moments_from_partials_f32(F, 0, 64, 1e-5f, F, F, F, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bn_nhwc_moments_from_partials_f32(nullptr, 4, 64, 1e-5f, F, F, F, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_bn_relu_nhwc_apply_f32(F, 64, F, F, F, F, F, nullptr, nullptr, nullptr, 0.1f, F, nullptr, 100, 66, 1, nullptr),
           NW_ERR_INVALID_ARG);                                                                 // c % 4
    EXPECT(nw_bn_relu_nhwc_apply_f32(F, 2564, F, F, F, F, F, nullptr, nullptr, nullptr, 0.1f, F, nullptr, 100, 2564, 1, nullptr),
           NW_ERR_UNSUPPORTED);                                                                 // c > 2560
    EXPECT(nw_bn_relu_nhwc_train_bwd_f32(F, 64, F, F, F, F, F, F, F, F, nullptr, 0, 32, nullptr, ws, sizeof ws, 100, 64, 1, nullptr),
           NW_ERR_INVALID_ARG);                                                                 // lddx < c
    // the pools
    unsigned char* TAP = reinterpret_cast<unsigned char*>(F);
    EXPECT(nw_avgpool2x2_nhwc_f32(F, 0, F, 0, 2, 8, 8, 6, nullptr), NW_ERR_INVALID_ARG);            // c % 4
    EXPECT(nw_avgpool2x2_nhwc_f32(F, 0, F, 0, 2, 1, 8, 8, nullptr), NW_ERR_INVALID_ARG);            // no 2 x 2 window
    EXPECT(nw_avgpool2x2_nhwc_f32(F, 4, F, 0, 2, 8, 8, 8, nullptr), NW_ERR_INVALID_ARG);            // ldx < c
    EXPECT(nw_avgpool2x2_nhwc_f32(nullptr, 0, F, 0, 2, 8, 8, 8, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_avgpool2x2_nhwc_f32(nullptr, 0, nullptr, 0, 0, 8, 8, 8, nullptr), NW_OK);            // empty batch
    EXPECT(nw_avgpool2x2_nhwc_bwd_f32(F, 0, F, 6, 2, 8, 8, 8, nullptr), NW_ERR_INVALID_ARG);        // ldgx < c
    EXPECT(nw_avgpool2x2_nhwc_bwd_f32(F, 0, nullptr, 0, 2, 8, 8, 8, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_maxpool3x3s2_nhwc_f32(F, 0, F, 0, TAP + 1, 2, 8, 8, 8, nullptr), NW_ERR_INVALID_ARG); // tap buffer not 4-byte aligned
    EXPECT(nw_maxpool3x3s2_nhwc_f32(F, 0, F, 10, TAP, 2, 8, 8, 8, nullptr), NW_ERR_INVALID_ARG);    // ldy % 4
    EXPECT(nw_maxpool3x3s2_nhwc_f32(F, 0, F, 0, TAP, 2, 0, 8, 8, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_maxpool3x3s2_nhwc_bwd_f32(F, 0, nullptr, F, 0, 2, 8, 8, 8, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_maxpool3x3s2_nhwc_bwd_f32(F, 0, TAP, F, 0, -1, 8, 8, 8, nullptr), NW_ERR_INVALID_ARG);
    // the optimizer step
    nw_sgd_param sp[2] = {{F, F, F, 64}, {F, F, nullptr, 64}};
    EXPECT(nw_sgd_step_f32(sp, 2, 0.1f, 0.9f, 1e-4f, 1, 0, nullptr), NW_ERR_INVALID_ARG);          // momentum without a buffer
    EXPECT(nw_sgd_step_f32(sp, 1, 0.1f, 0.0f, 1e-4f, 1, 0, nullptr), NW_ERR_INVALID_ARG);          // nesterov without momentum
    EXPECT(nw_sgd_step_f32(nullptr, 1, 0.1f, 0.9f, 0.f, 0, 0, nullptr), NW_ERR_INVALID_ARG);
    EXPECT(nw_sgd_step_f32(nullptr, 0, 0.1f, 0.9f, 0.f, 0, 0, nullptr), NW_OK);
    sp[0].n = -1;
    EXPECT(nw_sgd_step_f32(sp, 1, 0.1f, 0.9f, 0.f, 0, 0, nullptr), NW_ERR_INVALID_ARG);
    std::printf(failures ? "abi_args: %d FAILED\n" : "abi_args: all argument checks refused as documented\n", failures);
    return failures ? 1 : 0;
}
