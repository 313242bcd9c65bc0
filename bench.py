#!/usr/bin/env python3
"""bench.py -- NW-head hot path on MI355X.  Contract: see the task statement / DESIGN.md.

One "step" = one query batch (B=256, d=512) predicted against the whole support bank
('full' inference, nwhead/nw.py:127-160 with mode='full'): scores -> softmax -> label aggregation ->
log.  Workload "K3": bank N=50000, d=512, C=200 (BASELINE.json configs[2]); with --gpus G the bank is
sharded G ways (strong scaling), partials are exchanged with one RCCL all-gather per bucket of steps.
The north-star shape T (B=256, N=10000, d=512) is measured in the same run on rank 0 and reported in
the same JSON line under "north_star_T".
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 MFMA peak (= vector peak)
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
# The 'full'-inference fast path evaluates every fp32 multiply-add of the dot products with THREE fp16
# MFMA multiply-adds (x = h + l split, tile_f16.h), so its matrix-pipe roofline in ALGORITHMIC flops is
PEAK_SPLIT_F16_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
PEAK_HBM_GBS = 8000.0


def alg_bytes(B, N, d, C):
    return 4 * B * d + 4 * N * d + 8 * N + 4 * B * C


def alg_flops(B, N, d):
    return 2 * B * N * d + 10 * B * N


def make_inputs(B, N, d, C, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, d, generator=g)
    s = torch.randn(N, d, generator=g)
    sy = (torch.arange(N) % C).sort().values            # class-sorted, balanced like the 'full' bank
    return q.to(dev), s.to(dev), sy.to(dev)


PMC_FILE = "profiles/r04_bench_pmc.json"   # rocprofv3 --pmc passes of `bench.py --skip-extras` (tools/prof_bench.sh)
PMC_FILE_T = "profiles/r04_T_forward_pmc.json"   # ... of the T-shape forward (tools/prof.sh)


def pmc_entry(kernel, pmc_file=None):
    """The committed rocprofv3 PMC record of `kernel` (tools/save_profile.py: counters per full-size dispatch, the
    dispatch duration inside the counter passes, the clock and matrix-pipe occupancy derived from them), or None."""
    try:
        for k, v in json.load(open(os.path.join(ROOT, pmc_file or PMC_FILE))).items():
            if kernel in k:
                return v
    except Exception:
        pass
    return None


def pmc_traffic(kernel, pmc_file=None):
    """(HBM bytes per launch of `kernel`, source) from the committed rocprofv3 PMC passes of this very command
    (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE; counters cannot be collected inside a timed run), or
    (None, why).  The file belongs to one round: a stale profile is named, never silently mixed with fresh timings."""
    pmc_file = pmc_file or PMC_FILE
    v = pmc_entry(kernel, pmc_file)
    if v is None:
        return None, f"{pmc_file}: no entry for {kernel}"
    if "hbm_bytes_per_launch" not in v:
        return None, f"{pmc_file}: no HBM counters for {kernel}"
    return v["hbm_bytes_per_launch"], pmc_file


def time_kernel_events(fn, iters, warmup=3, min_warm_ms=30.0):
    """Average device time of fn() over `iters` launches on the current stream (HIP events), taken after
    `warmup` calls and at least `min_warm_ms` of back-to-back device time (the post-idle clock ramp:
    see the warm-up comment in main())."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if min_warm_ms > 0:
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        one = max(e0.elapsed_time(e1), 1e-3)
        for _ in range(min(int(min_warm_ms / one), 200) * iters):   # bounded: host-bound callers stay short
            fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3      # seconds


def measure_shape(B, N, d, C, dev, iters):
    """Single-GPU forward at one shape: whole-op time and the dominant (scores) kernel's time."""
    from nwhead_amd import ops
    q, s, sy = make_inputs(B, N, d, C, dev)
    cache = ops.SplitBank(s)       # what precompute() keeps for the bank: norms + split-fp16 rows
    t_fwd = time_kernel_events(lambda: ops.nw_head(q, s, sy, C, support_cache=cache), iters)
    t_n32 = time_kernel_events(lambda: ops.nw_head(q, s, sy, C, support_norm2=cache.norm2), iters)
    t_gen = time_kernel_events(lambda: ops.nw_head(q, s, sy, C), iters)
    t_sc = t_fwd
    fl = alg_flops(B, N, d)
    # the tile kernel alone (HIP events on its launch stream inside the library, nw_debug_tile_timing)
    import ctypes
    from nwhead_amd import _lib
    lib = _lib.load()
    lib.nw_debug_tile_timing(1)
    for _ in range(20):
        ops.nw_head(q, s, sy, C, support_cache=cache)
    tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    lib.nw_debug_tile_timing_read(ctypes.byref(tot), ctypes.byref(cnt))
    lib.nw_debug_tile_timing(0)
    t_tile = tot.value / max(cnt.value, 1) * 1e-6
    return {"B": B, "N": N, "d": d, "C": C, "ms_per_call": t_fwd * 1e3, "query_pred_per_s": B / t_fwd,
            "tile_kernel_us": t_tile * 1e6, "tile_kernel_us_note": "HIP-event pair around the tile kernel: ~5 us of event overhead at this size",
            "ms_per_call_fp32_mfma_cached_norms": t_n32 * 1e3,
            "ms_per_call_generic_forward_no_cache": t_gen * 1e3,
            "alg_GBps": alg_bytes(B, N, d, C) / t_fwd / 1e9, "frac_hbm": alg_bytes(B, N, d, C) / t_fwd / 1e9 / PEAK_HBM_GBS,
            "fwd_TFLOPs": 2 * B * N * d / t_sc / 1e12,
            "frac_of_fp32_mfma_peak": 2 * B * N * d / t_sc / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "frac_of_split_fp16_peak": 2 * B * N * d / t_sc / 1e12 / PEAK_SPLIT_F16_TFLOPS,
            "whole_op_frac_of_roofline": max(alg_bytes(B, N, d, C) / (PEAK_HBM_GBS * 1e9), fl / (PEAK_SPLIT_F16_TFLOPS * 1e12)) / t_fwd}


def measure_train_head(B, N, d, C, dev, iters=30):
    """A4 at a large shape: nll_loss(NWHead(x, sx, sy)).backward() with gradients for queries and supports (the
    supports split once for the forward and the backward, scores written by the forward, coefficients, the two
    products on the fp16 matrix cores).  Two figures: the Python-driven step (torch's autograd bookkeeping and ~15
    launches: bound by the host) and the same step captured once as a HIP graph (torch.cuda.CUDAGraph) and
    replayed: bound by the kernels."""
    import torch.nn.functional as F
    from nwhead_amd import ops
    q, s, sy = make_inputs(B, N, d, C, dev)
    q.requires_grad_(True)
    s.requires_grad_(True)
    t = torch.randint(0, C, (B,), device=dev)

    def step():
        q.grad = s.grad = None
        F.nll_loss(ops.nw_head(q, s, sy, C), t).backward()
    dt = time_kernel_events(step, iters)
    res = {"B": B, "N": N, "d": d, "C": C, "ms_per_fwd_bwd": dt * 1e3,
           "TFLOPs_fwd_plus_bwd_products": 6 * B * N * d / dt / 1e12}
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        gq_ref, gs_ref = q.grad.clone(), s.grad.clone()
        graph = torch.cuda.CUDAGraph()
        q.grad = s.grad = None
        with torch.cuda.graph(graph):
            F.nll_loss(ops.nw_head(q, s, sy, C), t).backward()
        graph.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(q.grad, gq_ref) and torch.equal(s.grad, gs_ref))
        dg = time_kernel_events(graph.replay, iters)
        res.update({"ms_per_fwd_bwd_hip_graph": dg * 1e3, "TFLOPs_fwd_plus_bwd_products_hip_graph": 6 * B * N * d / dg / 1e12,
                    "hip_graph_gradients_equal_eager": same})
    except Exception as e:   # a graph that cannot be captured is a finding, not a crash of the bench
        res["hip_graph_error"] = repr(e)[:200]
    return res


def measure_shuffled(B, N, d, C, dev, iters=20):
    """SURVEY 8d's shuffled-label variant of the K3 launch (B coalesced queries vs the whole bank): the tile
    kernels sum per run of equal labels, so the prepared bank keeps a class-sorted copy (ops.SplitBank(s, labels))."""
    from nwhead_amd import ops
    q, s, sy = make_inputs(B, N, d, C, dev)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(dev)
    sy_sh = sy[perm].contiguous()
    s_sh = s[perm].contiguous()
    bank_sorted, bank_sh = ops.SplitBank(s, sy), ops.SplitBank(s_sh, sy_sh)
    t_sorted = time_kernel_events(lambda: ops.nw_head(q, s, sy, C, support_cache=bank_sorted), iters)
    t_sh = time_kernel_events(lambda: ops.nw_head(q, s_sh, sy_sh, C, support_cache=bank_sh), iters)
    return {"B": B, "N": N, "ms_per_launch_sorted": t_sorted * 1e3, "ms_per_launch_shuffled": t_sh * 1e3,
            "query_pred_per_s_shuffled": B / t_sh}


def measure_influence(B, N, C, dev, iters=100):
    """K5: support_influence over a 10000-image support bank: a streaming kernel, 8*B*N + 8*N + 12*B algorithmic
    bytes per call.  Timed over a ROTATION of input / output sets whose total (> 600 MB) exceeds the
    256 MB Infinity Cache plus L2, so that every call streams from and to HBM (`frac_hbm`); the single-set figure,
    which lives in the caches between iterations, is reported next to it and labelled so."""
    from nwhead_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    nbytes = 8 * B * N + 8 * N + 12 * B
    nsets = max(2, int(640e6 // nbytes) + 1)
    sy = (torch.arange(N) % C).sort().values.to(dev)
    sets = []
    for k in range(nsets):
        w = torch.softmax(torch.randn(B, N, generator=g), -1).to(dev) if k < 4 else sets[k % 4][0].clone()
        probs = torch.softmax(torch.randn(B, C, generator=g), -1).to(dev)
        qy = torch.randint(0, C, (B,), generator=g).to(dev)
        sets.append((w, probs, qy, torch.empty(B, N, dtype=torch.float32, device=dev)))
    stream = torch.cuda.current_stream(dev).cuda_stream
    # the C ABI directly: at 20 MB the kernel is shorter than the Python wrapper's bookkeeping
    argl = [(p.data_ptr(), y.data_ptr(), w.data_ptr(), sy.data_ptr(), o.data_ptr(), B, N, C, stream) for w, p, y, o in sets]
    state = {"k": 0}

    def rotate():
        lib.nw_support_influence_f32(*argl[state["k"] % nsets])
        state["k"] += 1
    t_hbm = time_kernel_events(rotate, nsets * 2, warmup=nsets)
    t_one = time_kernel_events(lambda: lib.nw_support_influence_f32(*argl[0]), iters)
    roof = {"bound": "hbm", "kernel": "nw_influence_kernel<false>", "achieved": nbytes / t_hbm / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": nbytes / t_hbm / 1e9 / PEAK_HBM_GBS, "traffic": None,
            "traffic_source": "profiles/r04_K5_pmc_by_kernel.json not committed",
            "note": "achieved = algorithmic bytes / time per call in a stream of calls over the rotation (kernel boundaries included)"}
    try:
        e = json.load(open(os.path.join(ROOT, "profiles/r04_K5_pmc_by_kernel.json")))
        for k, v in e.items():
            if k.startswith("nw_influence_kernel<false>") and "hbm_read_MB_per_call_x2_corrected" in v:
                roof["traffic"] = (v["hbm_read_MB_per_call_x2_corrected"] + v["hbm_write_MB_per_call"]) * 1e6
                roof["traffic_source"] = ("profiles/r04_K5_pmc_by_kernel.json: FETCH_SIZE x 2 + WRITE_SIZE averaged over ALL stand-alone calls of "
                                          "tools/k5_time.py (~6 700 at B=256 = 20.6 MB algorithmic each, ~50 at B=4096 = 328 MB: 22.8 MB "
                                          "algorithmic on average)")
    except Exception:
        pass
    return {"B": B, "N": N, "C": C, "alg_bytes_per_call": nbytes, "roofline": roof if B == 256 else None,
            "us_per_call": t_hbm * 1e6, "alg_GBps": nbytes / t_hbm / 1e9, "frac_hbm": nbytes / t_hbm / 1e9 / PEAK_HBM_GBS,
            "rotation": f"{nsets} input/output sets, {nsets * nbytes / 1e6:.0f} MB in total: beyond the 256 MB Infinity Cache",
            "us_per_call_cache_resident": t_one * 1e6, "alg_GBps_cache_resident": nbytes / t_one / 1e9,
            "note_cache_resident": "one 20.6 MB set reused: served by L2 / Infinity Cache, NOT an HBM figure"}


def measure_forward_influence(B, N, d, C, dev, iters=100):
    """Forward + support_influence as ONE call (nw_fwd_influence_f32: scores written by the fused tile kernel, one
    in-place pass) against the two separate calls (forward with softmax weights, then the influence kernel)."""
    from nwhead_amd import ops
    q, s, sy = make_inputs(B, N, d, C, dev)
    qy = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(5)).to(dev)
    bank = ops.SplitBank(s, sy)
    t_fused = time_kernel_events(lambda: ops.nw_head_influence(q, s, sy, C, qy, support_cache=bank), iters)

    def two():
        out, w = ops.nw_head(q, s, sy, C, return_weights=True, support_cache=bank)
        return ops.support_influence_idx(out.exp(), qy, w, sy)
    t_two = time_kernel_events(two, iters)
    t_fwd = time_kernel_events(lambda: ops.nw_head(q, s, sy, C, support_cache=bank), iters)
    return {"B": B, "N": N, "d": d, "us_forward_plus_influence_one_call": t_fused * 1e6,
            "us_forward_with_weights_then_influence": t_two * 1e6, "us_forward_alone": t_fwd * 1e6}


def measure_latency(bank, q, iters=50):
    """Device latency of ONE predict('full') call of B queries against this rank's bank (no coalescing): HIP events
    around single, synchronised calls; median."""
    ts = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        bank.predict(q)
    for _ in range(iters):
        torch.cuda.synchronize()
        e0.record()
        bank.predict(q)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def backbone_roofline(gflop, seconds, pmc_json, launches_per_call):
    """`roofline` block of a backbone config in the head's convention: achieved = algorithmic fp32 flops / measured time, peak =
    the guide's dense fp16 MFMA peak / 3 (three fp16 products per fp32 multiply-add: OUR construction, as in `roofline`),
    frac_fp16_pipe the same number read as matrix-pipe utilisation; traffic (HBM-side bytes per call, FETCH_SIZE x 2 + WRITE_SIZE)
    and mfma_busy_frac (time-weighted over the convolution / weight-gradient kernels) from the committed rocprofv3 counter
    passes of the same workload (tools/prof_backbone_pmc.sh -> tools/pmc_by_kernel.py), which cannot run inside a timed region."""
    ach = gflop / seconds / 1e3
    blk = {"bound": "mfma", "kernel": "nw_conv_nhwc_kernel / nw_conv_wgrad_batch_kernel (whole backbone step)", "achieved": ach,
           "unit": "TFLOP/s", "peak": PEAK_SPLIT_F16_TFLOPS, "frac": ach / PEAK_SPLIT_F16_TFLOPS,
           "frac_fp16_pipe": 3 * ach / PEAK_F16_MFMA_TFLOPS, "operands": "split-fp16x2",
           "traffic": None, "traffic_source": f"{pmc_json} not committed", "mfma_busy_frac": None}
    try:
        d = json.load(open(os.path.join(ROOT, pmc_json)))
        conv = {k: v for k, v in d.items() if k.startswith("nw_conv_nhwc_kernel") or k.startswith("nw_conv_wgrad")}
        us = sum(v["device_us_total"] for v in conv.values())
        calls = launches_per_call
        tot = sum((v.get("hbm_read_MB_per_call_x2_corrected", 0.0) + v.get("hbm_write_MB_per_call", 0.0)) * v["calls"] for v in d.values())
        blk["traffic"] = tot * 1e6 / calls
        blk["traffic_source"] = pmc_json + f" (all kernels of {calls} profiled calls; FETCH_SIZE x 2 + WRITE_SIZE)"
        blk["mfma_busy_frac"] = sum(v.get("mfma_busy_frac", 0.0) * v["device_us_total"] for v in conv.values()) / us if us else None
        blk["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), time-weighted over the convolution and weight-gradient kernels"
        blk["conv_share_of_device_time"] = us / sum(v["device_us_total"] for v in d.values())
    except Exception:
        pass
    return blk


def measure_backbone_configs(dev):
    """BASELINE configs[1] and [3] end to end: K2 = ResNet-18 + head, predict over 64 images @224 against a 1000-row
    bank (plain: torch / MIOpen; what NWNet.predict runs after enable_bn_folding: the channels_last copy whose every
    convolution is nw_conv2d_nhwc_f16x2); K4 = DenseNet-121 training step (joint forward of 32 queries + 10 supports
    @224 on the channels-last path: own convolutions forward / data / weight gradient, own BatchNorm + ReLU; NLL loss,
    backward through the HIP head, SGD), with NW_NHWC_TRAINING-off (NCHW, MIOpen convolutions) beside it."""
    import torch.nn.functional as F
    from nwhead_amd.model import load_model
    from nwhead_amd.nwhead.kernel import get_kernel
    from nwhead_amd.nwhead.nw import NWHead
    out = {}
    g = torch.Generator().manual_seed(7)
    try:
        net = load_model("resnet18").to(dev).eval()
        x = torch.randn(64, 3, 224, 224, generator=g).to(dev)
        s = torch.randn(1000, 512, generator=g).to(dev)
        sy = (torch.arange(1000) % 200).sort().values.to(dev)
        head = NWHead(get_kernel("euclidean"), 200)

        def k2():
            with torch.no_grad():
                return head(net(x), s, sy)
        t = time_kernel_events(k2, 10)
        # what NWNet.predict runs after precompute() with enable_bn_folding(True, channels_last=True)
        from nwhead_amd.model import fold_batchnorm
        folded = fold_batchnorm(net).to(memory_format=torch.channels_last)
        xcl = x.contiguous(memory_format=torch.channels_last)

        def k2f():
            with torch.no_grad():
                return head(folded(xcl), s, sy)
        tf = time_kernel_events(k2f, 10)
        gf = 3.63 * 64                                        # GFLOP, ResNet-18 forward @224
        out["config_K2_resnet18_plus_head"] = {"images": 64, "N": 1000, "ms_per_call": tf * 1e3, "images_per_s": 64 / tf,
                                               "backbone": "BatchNorm-folded channels_last copy (what NWNet.predict runs): "
                                                           "nw_conv2d_nhwc_f16x2 (split-fp16 MFMA, fp32-grade)",
                                               "backbone_TFLOPs": gf / tf / 1e3,
                                               "roofline": backbone_roofline(gf, tf, "profiles/r04_K2_pmc_by_kernel.json", 10),
                                               "ms_per_call_plain_torch_miopen": t * 1e3,
                                               "note_plain": "the unfolded eval-mode network on torch / MIOpen fp32, for comparison"}
        del net, folded
    except Exception as e:                                    # never lose the JSON line to an extra
        out["config_K2_resnet18_plus_head"] = {"error": repr(e)[:200]}
    try:
        dn = load_model("densenet121").to(dev).train()
        from nwhead_amd.optim import SGD as _NWSGD
        opt = _NWSGD(dn.parameters(), lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4)
        xq = torch.randn(32, 3, 224, 224, generator=g).to(dev)
        yq = torch.randint(0, 10, (32,), generator=g).to(dev)
        xs = torch.randn(10, 3, 224, 224, generator=g).to(dev)
        ys = torch.arange(10).to(dev)
        head = NWHead(get_kernel("euclidean"), 10)

        def k4():
            opt.zero_grad(set_to_none=True)
            feats = dn(torch.cat((xq, xs)))
            loss = F.nll_loss(head(feats[:32], feats[32:], ys), yq)
            loss.backward()
            opt.step()
        # one untimed rehearsal of the timed call (first-step effects: the stem's weight-gradient set-up, allocator growth) and the
        # interpreter's collections taken now, not inside a run (BENCH_r03 had one 24.9 ms run among 15.4 ms ones)
        time_kernel_events(k4, 8, warmup=10)
        import gc
        gc.collect()
        runs4 = sorted(time_kernel_events(k4, 8, warmup=2) for r in range(5))
        t = runs4[2]                                          # median of five runs of 8 steps (a shared host is noisy)
        gf4 = 5.67 * 3 * 42                                   # GFLOP, DenseNet-121 fwd+bwd over 32 + 10 images @224
        import nwhead_amd.model.backbones as BB
        t_nchw = None
        if BB.NHWC_TRAINING:                                  # the NCHW path (MIOpen convolutions) beside it
            BB.NHWC_TRAINING = False
            try:
                t_nchw = time_kernel_events(k4, 5, warmup=10)   # MIOpen's first calls of a configuration run slow stand-in kernels
            finally:
                BB.NHWC_TRAINING = True
        out["config_K4_densenet121_train_step"] = {"B": 32, "n_way": 10, "n_shot": 1, "ms_per_step": t * 1e3,
                                                   "ms_per_step_runs": [round(r * 1e3, 3) for r in runs4],
                                                   "backbone": "channels-last: nw_conv2d_nhwc_f16x2 / nw_conv2d_nhwc_wgrad_f16x2 "
                                                               "(split-fp16 MFMA, fp32-grade) + nw_bn_relu_nhwc_train_*"
                                                               if BB.NHWC_TRAINING else "torch/MIOpen fp32 (NCHW)",
                                                   "backbone_TFLOPs": gf4 / t / 1e3,
                                                   "roofline": backbone_roofline(gf4, t, "profiles/r04_K4_pmc_by_kernel.json", 6),
                                                   "ms_per_step_nchw_miopen": None if t_nchw is None else t_nchw * 1e3}
        # the inference side of the same backbone (precompute / predict): plain eval vs the folded copy whose
        # BatchNorm -> ReLU pairs run in nw_scale_shift_relu_f32
        from nwhead_amd.model import fold_batchnorm
        dn.eval()
        del opt
        x64 = torch.randn(64, 3, 224, 224, generator=g).to(dev)
        with torch.no_grad():
            folded = fold_batchnorm(dn)
            t_plain = time_kernel_events(lambda: dn(x64), 5, warmup=3)
            t_fold = time_kernel_events(lambda: folded(x64), 5, warmup=3)
        out["densenet121_eval_forward"] = {"images": 64, "ms_plain": t_plain * 1e3, "ms_folded": t_fold * 1e3,
                                           "images_per_s_folded": 64 / t_fold,
                                           "note": "round 4: both run DenseNet._forward_nhwc_infer (channels-last split-fp16 kernels, "
                                                   "BatchNorms inside the convolutions); round 3: 11.0 / 5.1 ms",
                                           "TFLOPs": 5.67 * 64 / t_fold / 1e3,
                                           "frac_split_fp16": 5.67 * 64 / t_fold / 1e3 / PEAK_SPLIT_F16_TFLOPS}
    except Exception as e:
        out.setdefault("config_K4_densenet121_train_step", {"error": repr(e)[:200]})
        out.setdefault("densenet121_eval_forward", {"error": repr(e)[:200]})
    return out


def cpu_baseline(B_sample, N, d, C, budget_s=20.0):
    """The reference's op sequence (oracle port) on this host's cores, bounded sample."""
    from oracle import nw_oracle as O
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B_sample, d, generator=g)
    s = torch.randn(N, d, generator=g)
    sy = (torch.arange(N) % C).sort().values
    # threads actually usable: the scheduler affinity of this process, capped at the GPU box's CPU
    # share for one GPU (16); os.cpu_count() reports every core of the host (256 on the MI355X node)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("NW_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    with torch.no_grad():
        O.nw_head_f32(q, s, sy, C)                          # warm-up
        times, t_all = [], time.perf_counter()
        while len(times) < 5 and (time.perf_counter() - t_all) < budget_s:
            t0 = time.perf_counter()
            O.nw_head_f32(q, s, sy, C)
            times.append(time.perf_counter() - t0)
        # SURVEY 8d also asks for the single-thread figure: 8 queries, two calls
        torch.set_num_threads(1)
        q1 = q[:8]
        O.nw_head_f32(q1, s, sy, C)
        t0 = time.perf_counter()
        O.nw_head_f32(q1, s, sy, C)
        t1 = time.perf_counter() - t0
        torch.set_num_threads(cores)
    times.sort()
    med = times[len(times) // 2]
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), model)
    except OSError:
        pass
    return {"value": B_sample / med, "unit": "query-predictions/s", "cores": cores, "kind": "port",
            "sample": f"B={B_sample} queries x full bank N={N}, d={d}, C={C}; median of {len(times)} calls, "
                      f"torch {torch.__version__} CPU fp32, {cores} threads",
            "value_1_thread": 8 / t1, "sample_1_thread": f"B=8 queries x the same bank, one call after one warm-up",
            "cpu_model": model, "host_cores_total": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--bank", type=int, default=50000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--classes", type=int, default=200)
    ap.add_argument("--bucket", type=int, default=26,
                    help="query batches coalesced per launch (and per RCCL all-gather); the same at every --gpus, so "
                         "that the scaling curve compares like with like.  26: the launch's tile count (52 query "
                         "tiles x 391 / 196 / 98 / 49 support tiles at 1 / 2 / 4 / 8 shards of the 50000-row bank) is "
                         "within 0.7 %% below a multiple of the 256 CUs at every one of them; at 16 the eight-shard "
                         "launch is 6.1 tiles per CU, i.e. 7 rounds for the work of 6.1")
    ap.add_argument("--min-warmup-ms", type=float, default=40.0,
                    help="the untimed warm-up lasts at least this long (device time): post-idle clock ramp")
    ap.add_argument("--min-timed-ms", type=float, default=100.0, help="the timed region lasts at least this long")
    ap.add_argument("--min-launches", type=int, default=100, help="... and at least this many coalesced launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-extras", action="store_true", help="only the timed workload (for profiling)")
    args = ap.parse_args()
    args.bucket = max(1, args.bucket)
    # The timed region is made of WHOLE launches and of at least four of them: --steps is rounded up to a multiple
    # of the bucket (a ragged last bucket would be a launch shape of its own, and two launches are not a
    # measurement); `steps` in the JSON line is what was timed, `steps_requested` what was asked for.
    steps_requested = args.steps
    args.steps = max(4 * args.bucket, -(-args.steps // args.bucket) * args.bucket)
    # ... --steps is a MINIMUM: the timed region also lasts at least --min-timed-ms and --min-launches coalesced launches
    # (fixed below, once a step's duration is known); `steps` in the line is what was timed
    # Library banners (RCCL prints its version block to stdout when the first communicator comes up)
    # must not land next to the JSON line: stdout is pointed at stderr until the line is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # Rehearsal knobs for a one-GPU box (never set by the driver): NW_BENCH_ONE_DEVICE=1 puts every rank on
    # cuda:0 and NW_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device), so that the
    # multi-rank code path -- shards, class windows, all-gather, merge of G partials -- runs the real kernels.
    if os.environ.get("NW_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # NW_FORCE_DIST=1 rehearses the RCCL code path on a single rank (the 1-GPU box cannot host two)
    use_dist = world > 1 or os.environ.get("NW_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("NW_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from nwhead_amd import ops
    from nwhead_amd.sharded import ShardedBank, shard_bounds

    B, N, d, C = args.batch, args.bank, args.dim, args.classes
    q, s, sy = make_inputs(B, N, d, C, dev)
    lo, hi = shard_bounds(N, world, rank)
    bank = ShardedBank(s[lo:hi].clone(), sy[lo:hi].clone(), C)
    del s
    if use_dist:   # every rank must ship the same row layout: one class-window width for all (ShardedBank's all-gather)
        cls = torch.tensor([bank.CL], dtype=torch.int64, device=dev)
        allc = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, cls)
        assert bool((allc == bank.CL).all()), f"class-window widths differ across ranks: {allc.tolist()}"
    backend_used = (os.environ.get("NW_DIST_BACKEND", "nccl") if use_dist else None)
    # one bucket of distinct query batches, back to back in one staging buffer (predict_stream then hands the
    # bucket to the kernels as a view instead of concatenating it)
    gq = torch.Generator().manual_seed(123)
    nq = max(args.bucket, 4)
    qbuf = torch.randn(nq * B, d, generator=gq).to(dev)
    qs = [qbuf[k * B:(k + 1) * B] for k in range(nq)]

    def run(nsteps):
        # same code path at every N: buckets of query batches coalesced per launch (sharded.py); long
        # (warm-up) runs go in chunks of 32 buckets so that the list of outputs stays small
        chunk = 32 * args.bucket
        out = None
        for i0 in range(0, nsteps, chunk):
            n = min(chunk, nsteps - i0)
            out = bank.predict_stream([qs[i % nq] for i in range(n)], bucket=args.bucket)[-1]
        return out

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed warm-up: at least W steps, and at least one bucket of every launch shape the timed region
    # will use (a full bucket, and the shorter last one), so that workspace growth, the packed/gathered
    # ring buffers and RCCL's first collective of each size stay outside the timed region
    warm_steps = max(args.warmup, (3 if use_dist else 1) * args.bucket)   # 3: predict_stream's ring of exchange buffers
    run(args.bucket)
    # The interpreter's first FULL garbage collection walks everything torch imported: a 40-70 ms host pause that
    # landed inside the timed region in about one run out of five (tools/stall_probe.py: launches 150-200 of a
    # process).  Like a serving process after start-up: collect once everything is set up, then move the survivors
    # out of the collector's sight (the young generations keep running) -- here, ahead of the warm-up, so that the
    # pause is not an idle period right in front of the timed region either.
    import gc
    gc.collect()
    gc.freeze()
    run(warm_steps)
    if args.steps % args.bucket:
        run(args.steps % args.bucket)
        warm_steps += args.steps % args.bucket
    # ... and until the device has left its post-idle ramp: measured on MI355X (tools/bucket_times.py),
    # back-to-back launches of this workload take 886, 684, ... 775 ... 680 (12th) ... 650 (25th) ... 625 us
    # (40th and on) after an idle period; a serving process sits in the steady state, so the untimed
    # warm-up runs for at least --min-warmup-ms of device time (the same number of steps on every rank)
    per_step_ms = 0.05
    if args.min_warmup_ms > 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(8 * args.bucket)
        e1.record()
        torch.cuda.synchronize()
        per_step_ms = max(e0.elapsed_time(e1) / (8 * args.bucket), 1e-6)
        extra = int(max(0.0, args.min_warmup_ms - e0.elapsed_time(e1)) / per_step_ms / args.bucket + 1) * args.bucket
        if use_dist:                                   # one count for all ranks (collectives inside)
            cnt = torch.tensor([extra], dtype=torch.int64, device=dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
            extra = int(cnt.item())
        run(extra)
        warm_steps += 8 * args.bucket + extra
    warm_steps += args.bucket   # (the bucket in front of the collection)
    # the timed region: at least --steps steps, --min-launches whole launches and --min-timed-ms of device time
    want = max(args.steps, args.min_launches * args.bucket, int(args.min_timed_ms / per_step_ms) + 1)
    want = -(-want // args.bucket) * args.bucket
    if use_dist:                                       # one count for all ranks
        cnt = torch.tensor([want], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        want = int(cnt.item())
    args.steps = want
    # ... and one untimed rehearsal of the timed call itself (same number of steps, same chunking): whatever the
    # caching allocator or the exchange ring still has to grow for THIS pattern grows here (measured: a 30 ms stall
    # inside an 8.7 ms timed region when the first 32-bucket chunk of a process was the timed one)
    if args.steps * per_step_ms < 2000.0:
        run(args.steps)
        warm_steps += args.steps
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()

    lat_us = measure_latency(bank, qs[0])     # every rank: the sharded call has a collective inside
    line = None
    if rank == 0:
        n_shard = hi - lo
        # one launch = one forward over this rank's shard = nw_fused_kernel (+ the small merge kernel)
        # one launch = the partial forward of one bucket (bucket*B queries) over this rank's shard:
        # nw_fused_kernel + nw_merge_runs_kernel
        Bl = B * args.bucket
        qcat = torch.cat([qs[i % nq] for i in range(args.bucket)], dim=0)
        pk = torch.empty(bank.row_len(Bl), dtype=torch.float32, device=dev)
        # the tile kernel alone, bracketed by HIP events on its own launch stream inside the library
        # (nw_debug_tile_timing, include/nwhead_hip.h), and the whole partial forward around it
        import ctypes
        from nwhead_amd import _lib
        lib = _lib.load()
        time_kernel_events(lambda: bank._partial(pk, qcat), 10)            # (device stays in its steady state)
        lib.nw_debug_tile_timing(1)
        t_all = time_kernel_events(lambda: bank._partial(pk, qcat), 20, warmup=0, min_warm_ms=0)
        tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        _lib.check(lib.nw_debug_tile_timing_read(ctypes.byref(tot), ctypes.byref(cnt)), "nw_debug_tile_timing_read")
        lib.nw_debug_tile_timing(0)
        t_sc = tot.value / max(cnt.value, 1) * 1e-6
        flops = 2.0 * Bl * n_shard * d
        fast = bank.cache is not None and bank.cache.split is not None
        peak = PEAK_SPLIT_F16_TFLOPS if fast else PEAK_F32_MFMA_TFLOPS
        persistent = fast and Bl * n_shard >= 64 * 128 * 1024
        kname = "nw_fused_f16p_kernel" if persistent else "nw_fused_kernel"
        traffic, traffic_src = pmc_traffic(kname)
        pmc = pmc_entry(kname) or {}
        ach = flops / t_sc / 1e12
        roof = {"bound": "mfma",
                "kernel": kname,
                "achieved": ach,
                "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                "operands": "split-fp16x2" if fast else "f32",
                "peak_note": ("`peak` is OUR construction, not a guide number: the guide's fp16 dense MFMA peak (2500 "
                              "TFLOP/s) divided by the 3 fp16 products issued per algorithmic fp32 multiply-add "
                              "(x = h + l split, al*bl dropped).  frac_fp16_pipe = 3 * achieved / 2500 (the same "
                              "number: matrix-pipe utilisation counting issued products); alg_frac_fp16_peak = "
                              "achieved / 2500 (algorithmic flops against the guide's peak, no factor 3); "
                              "achieved_vs_fp32_mfma_peak = achieved / 157.3 (the guide's fp32 MFMA peak, > 1 because "
                              "the work is not on the fp32 pipe).  clock_GHz / mfma_busy_frac (from the committed PMC passes of this "
                              "command) say at what clock and matrix-pipe occupancy it was reached: the kernel runs at "
                              "1.9-2.2 GHz (box to box) with the pipe busy about half of the time (DESIGN.md 4.3)"
                              if fast else "fp32 dense MFMA peak (guide)"),
                "frac_fp16_pipe": (3 * ach / PEAK_F16_MFMA_TFLOPS) if fast else None,
                "alg_frac_fp16_peak": ach / PEAK_F16_MFMA_TFLOPS,
                "achieved_vs_fp32_mfma_peak": flops / t_sc / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "kernel_us": t_sc * 1e6, "kernel_launches_timed": cnt.value,
                # from the committed counter passes of this command (cannot be collected inside a timed run): the shader
                # clock the kernel's waves saw, the share of their cycles the matrix pipe was busy, the dispatch duration
                # in those passes; an s_memtime / s_memrealtime diagnostic build of the same launch agrees
                "clock_GHz": pmc.get("clock_GHz_from_wave_cycles"), "clock_GHz_grbm": pmc.get("clock_GHz_from_grbm"),
                "mfma_busy_frac": pmc.get("mfma_busy_frac_of_wave_cycles"),
                "kernel_us_in_pmc_passes": pmc.get("duration_us_in_pmc_passes"), "clock_source": PMC_FILE,
                "launch_us": t_all * 1e6,
                "launch_note": "launch_us = query split + run tables + tile kernel + run merge (one partial forward)",
                "alg_flops_per_launch": flops,
                "queries_per_launch": Bl,
                "alg_bytes_per_launch": alg_bytes(Bl, n_shard, d, C),
                "alg_GBps": alg_bytes(Bl, n_shard, d, C) / t_all / 1e9}
        line = {"metric": "query-predictions/sec", "value": args.steps * B / dt, "unit": "query-predictions/s",
                "n_gpus": world, "steps": args.steps, "steps_requested": steps_requested, "warmup": args.warmup,
                "rccl_ranks": world if backend_used == "nccl" else 0, "dist_backend": backend_used,
                "class_window": bank.CL, "persistent_wgs": bank.persistent_wgs,
                "warmup_steps_run": warm_steps,
                "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "operands": "split-fp16x2 (fp32 accumulate)" if fast else "f32",
                "data": "synthetic",
                "latency_us_single_call_B256": lat_us,
                "latency_note": "one uncoalesced predict('full') call of B queries against this rank's shard (plus the "
                                "exchange when sharded); a `step` of the timed region is 1/bucket of a coalesced launch",
                "config": {"workload": f"K3 predict('full'): B={B} queries/step vs bank N={N} d={d} C={C}, "
                                       f"bank sharded {world}-way, {args.bucket} query batches coalesced per launch"
                                       + (" and per RCCL all-gather" if world > 1 else ""),
                           "B": B, "N_support": N, "d": d, "C": C, "parallelism": f"support-shard x{world}"},
                "roofline": roof}
        if world == 1 and not args.skip_extras:
            line["north_star_T"] = T_ = measure_shape(256, 10000, 512, 200, dev, 100)
            # the north-star shape as a roofline block of its own: the tile kernel at T
            kT = "nw_fused_kernel"
            trT, trT_src = pmc_traffic(kT, PMC_FILE_T)
            flT = 2.0 * 256 * 10000 * 512
            # the kernel's own duration: from the committed rocprofv3 kernel trace of this shape (an event pair around a
            # 17 us kernel adds ~5 us of its own: `tile_kernel_us_events` in north_star_T); live figure: the whole op
            # kernel_us is LIVE: HIP events around this run's launches, which read ~5 us high on a 15 us kernel; the
            # committed kernel trace's figure is carried beside it, labelled, and `achieved` / `frac` follow the live one
            kus, kus_src = T_["tile_kernel_us"], "this run: HIP events around the launch (they add ~5 us of their own to a 15 us kernel)"
            kus_prof, kus_prof_src = None, None
            try:
                import csv
                for name in ("profiles/r04_T_forward_stats.csv", "profiles/r03_T_forward_stats.csv"):
                    if not os.path.exists(os.path.join(ROOT, name)):
                        continue
                    for r in csv.DictReader(open(os.path.join(ROOT, name))):
                        if r["kernel"].startswith(kT + "<"):
                            kus_prof, kus_prof_src = float(r["avg_us"]), name + " (rocprofv3 --kernel-trace --stats)"
                            break
                    if kus_prof is not None:
                        break
            except Exception:
                pass
            achT = flT / (kus * 1e-6) / 1e12
            line["roofline_T"] = {"bound": "mfma", "kernel": kT, "kernel_us": kus, "kernel_us_source": kus_src,
                                  "kernel_us_committed_profile": kus_prof, "kernel_us_committed_profile_source": kus_prof_src,
                                  "frac_from_committed_profile": None if kus_prof is None else flT / (kus_prof * 1e-6) / 1e12 / PEAK_SPLIT_F16_TFLOPS,
                                  "whole_op_us": T_["ms_per_call"] * 1e3,
                                  "achieved": achT, "unit": "TFLOP/s", "peak": PEAK_SPLIT_F16_TFLOPS, "frac": achT / PEAK_SPLIT_F16_TFLOPS,
                                  "frac_vs_fp32_mfma_bound": achT / PEAK_F32_MFMA_TFLOPS,
                                  "whole_op_frac_split_fp16": flT / (T_["ms_per_call"] * 1e-3) / 1e12 / PEAK_SPLIT_F16_TFLOPS,
                                  "whole_op_frac_vs_fp32_mfma_bound": flT / (T_["ms_per_call"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                  "alg_bytes": alg_bytes(256, 10000, 512, 200), "alg_flops": flT,
                                  "traffic": trT, "traffic_source": trT_src, "operands": "split-fp16x2",
                                  "peak_note": "peak = 2500 / 3 TFLOP/s as in `roofline` (three fp16 products per fp32 multiply-add); "
                                               "frac_vs_fp32_mfma_bound is against SURVEY 8d's fp32 compute bound (16.7 us at T)"}
            line["config_K2_head"] = measure_shape(64, 1000, 512, 200, dev, 100)
            line["config_K5_support_influence"] = measure_influence(256, 10000, 200, dev)
            line["config_K5_support_influence"]["B4096"] = {k: v for k, v in measure_influence(4096, 10000, 200, dev, iters=20).items()
                                                             if k in ("B", "alg_bytes_per_call", "us_per_call", "alg_GBps", "frac_hbm", "rotation")}
            line["config_K5_support_influence"]["fused_with_forward_T"] = measure_forward_influence(256, 10000, 512, 200, dev)
            line["head_train_step_T"] = measure_train_head(256, 10000, 512, 200, dev)
            line["K3_shuffled_labels"] = measure_shuffled(4096, N, d, C, dev)
            line.update(measure_backbone_configs(dev))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(32, N, d, C)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
