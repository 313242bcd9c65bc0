#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<TAG>/ rocprofv3 output tree into profiles/<NAME>_{stats.csv,pmc.json,summary.md}.
usage: python tools/save_profile.py gpurun_out/prof_bench_r01_v2 r01_bench "command that was profiled"
"""
import collections, csv, glob, json, os, re, sys
src, name = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
os.makedirs("profiles", exist_ok=True)
def kname(full):
    m = re.search(r"(nw_[a-z_0-9]+(<[^>]*>)?)", full)
    return m.group(1) if m else full[:80]
def newest(pattern):
    """gpurun merges a run's files INTO the directory of an earlier run with the same tag: per leaf directory of `src`
    (trace, pmc1, pmc2, ...) only the most recent file counts."""
    by_dir = collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        by_dir[os.path.relpath(f, src).split(os.sep)[0]].append(f)
    return [max(fs, key=os.path.getmtime) for fs in by_dir.values()]


stats = []
for f in newest(src + "/trace/**/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        stats.append({"kernel": kname(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                      "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])})
stats.sort(key=lambda r: -r["pct"])
with open(f"profiles/{name}_stats.csv", "w") as fh:
    w = csv.DictWriter(fh, fieldnames=list(stats[0].keys())); w.writeheader(); w.writerows(stats)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)     # kernel -> {(pass file, dispatch id): microseconds}: the dispatch durations INSIDE the PMC passes
for f in newest(src + "/pmc*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[kname(r["Kernel_Name"])].setdefault(r["Counter_Name"], []).append(
                (float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
def full_size_mean(v):
    """Mean over the dispatches of the LARGEST launch shape (values within 10 % of the maximum): a run mixes launch
    shapes of one kernel (warm-up, the single-call latency probe, the coalesced launches of the timed region)."""
    top = max(v)
    big = [x for x in v if x >= 0.9 * top] if top > 0 else v
    return sum(big) / len(big)


pmc = {k: {c: full_size_mean(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in acc.items():
    pmc[k]["dispatches_all_shapes"] = max(len(v) for v in d.values())
for k, d in pmc.items():
    if dur.get(k):
        # duration of the full-size dispatches in the counter passes themselves, and what the counters say about the clock
        # (VERDICT r02 item 2): a persistent kernel's waves live as long as the kernel, so SQ_WAVE_CYCLES (quad-cycles)
        # x 4 / SQ_WAVES / duration is the shader clock they saw; GRBM_GUI_ACTIVE sums the 8 XCDs; the matrix pipe of
        # each of the 1024 SIMDs is busy SQ_VALU_MFMA_BUSY_CYCLES / 1024 cycles
        # median duration of the full-size dispatches (those whose counter value is within 10 % of the largest) of one counter
        cname = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in dur[k] else sorted(dur[k])[0]
        top = max(v for v, _ in dur[k][cname])
        ds = sorted(t_ for v, t_ in dur[k][cname] if top <= 0 or v >= 0.9 * top)
        d["duration_us_in_pmc_passes"] = ds[len(ds) // 2]
        t = d["duration_us_in_pmc_passes"] * 1e-6
        if d.get("SQ_WAVE_CYCLES") and d.get("SQ_WAVES"):
            d["cycles_per_wave"] = 4.0 * d["SQ_WAVE_CYCLES"] / d["SQ_WAVES"]
            d["clock_GHz_from_wave_cycles"] = d["cycles_per_wave"] / t / 1e9
            if d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                d["mfma_busy_frac_of_wave_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / d["cycles_per_wave"]
        if d.get("GRBM_GUI_ACTIVE"):
            d["clock_GHz_from_grbm"] = d["GRBM_GUI_ACTIVE"] / 8.0 / t / 1e9
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        # MI355X_MICROARCH.md, HBM section: FETCH_SIZE (KB) reports exactly half of a wide coalesced
        # read stream on gfx950 -> doubled; WRITE_SIZE (KB) is exact for 16-B streaming stores.
        d["hbm_read_bytes_corrected"] = 2 * d.get("FETCH_SIZE", 0.0) * 1024
        d["hbm_write_bytes"] = d.get("WRITE_SIZE", 0.0) * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
json.dump(pmc, open(f"profiles/{name}_pmc.json", "w"), indent=1, sort_keys=True)
with open(f"profiles/{name}_summary.md", "w") as fh:
    fh.write(f"# rocprofv3 summary `{name}`\n\ncommand: `{cmd}`\n\n## --kernel-trace --stats\n\n| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for r in stats:
        fh.write(f"| `{r['kernel']}` | {r['calls']} | {r['avg_us']:.2f} | {r['min_us']:.2f} | {r['max_us']:.2f} | {r['pct']:.1f} |\n")
    # the dominant kernel launch by launch: after an idle period the clocks ramp for ~40 launches,
    # so the all-launch average above sits above the steady state bench.py's timed region runs in
    for f in newest(src + "/trace/**/*kernel_trace.csv"):
        rows = [r for r in csv.DictReader(open(f)) if kname(r["Kernel_Name"]) == stats[0]["kernel"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        if len(d) >= 40:
            tail = d[-20:]
            fh.write(f"\n`{stats[0]['kernel']}` launch by launch: first 5 = {', '.join(f'{x:.0f}' for x in d[:5])} us; "
                     f"last 20 average {sum(tail) / len(tail):.1f} us (min {min(tail):.1f}, max {max(tail):.1f}) -- "
                     f"the steady state `roofline.kernel_us` of the bench line is measured in.\n")
    fh.write("\n## --pmc (separate passes; mean per dispatch of the kernel's largest launch shape)\n\n")
    for k, d in pmc.items():
        fh.write(f"### `{k}`\n\n| counter | mean per full-size dispatch |\n|---|---|\n")
        for c, v in sorted(d.items()):
            fh.write(f"| {c} | {v:,.1f} |\n")
        fh.write("\n")
print(open(f"profiles/{name}_summary.md").read()[:1500])
