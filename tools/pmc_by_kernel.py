#!/usr/bin/env python3
"""Per kernel (full template instantiation) totals of a tools/prof_backbone_pmc.sh output tree: device time and share from the
kernel trace; from the counter passes, summed over ALL dispatches of the kernel: matrix-pipe occupancy
(SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)), LDS bank-conflict share of the LDS-active cycles, and
HBM-side bytes (FETCH_SIZE x 2 on gfx950 for wide streaming reads -- MI355X_MICROARCH.md, HBM section -- + WRITE_SIZE).
usage: python tools/pmc_by_kernel.py gpurun_out/prof_TAG out.json"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, out = sys.argv[1], sys.argv[2]


def kname(full):
    m = re.search(r"(nw_[A-Za-z_0-9]+(<[^>]*>)?)", full)
    return m.group(1) if m else full[:70]


def newest(pattern):
    by_dir = collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        by_dir[os.path.relpath(f, src).split(os.sep)[0]].append(f)
    return [max(fs, key=os.path.getmtime) for fs in by_dir.values()]


trace = collections.defaultdict(lambda: [0, 0.0])
for f in newest(src + "/trace/**/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        trace[k][0] += 1
        trace[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
total_us = sum(v[1] for v in trace.values()) or 1.0
sums = collections.defaultdict(lambda: collections.defaultdict(float))
for f in newest(src + "/pmc*/**/*counter_collection.csv"):
    pas = os.path.relpath(f, src).split(os.sep)[0]
    for r in csv.DictReader(open(f)):
        k, c = kname(r["Kernel_Name"]), r["Counter_Name"]
        if c == "GRBM_GUI_ACTIVE":
            c = f"GRBM_GUI_ACTIVE@{pas}"
        sums[k][c] += float(r["Counter_Value"])
res = {}
for k, (calls, us) in sorted(trace.items(), key=lambda kv: -kv[1][1]):
    d = dict(sums.get(k, {}))
    e = {"calls": calls, "device_us_total": round(us, 1), "pct_of_device_time": round(100 * us / total_us, 2),
         "avg_us": round(us / calls, 2)}
    g1, g2 = d.get("GRBM_GUI_ACTIVE@pmc1"), d.get("GRBM_GUI_ACTIVE@pmc2")
    if g1 and d.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        e["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * g1 / 8.0), 4)
    if g1 and d.get("SQ_BUSY_CYCLES") is not None:
        e["sq_busy_cycles"] = d["SQ_BUSY_CYCLES"]
    if d.get("SQ_WAVE_CYCLES"):
        e["wait_any_frac_of_wave_cycles"] = round(d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 4)
        e["active_inst_frac_of_wave_cycles"] = round(d.get("SQ_ACTIVE_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 4)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac_of_lds_active"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
        if g2:
            e["lds_active_frac"] = round(d["SQ_LDS_IDX_ACTIVE"] / (256.0 * g2 / 8.0), 4)
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        rd, wr = 2 * d.get("FETCH_SIZE", 0.0) * 1024, d.get("WRITE_SIZE", 0.0) * 1024
        n = max(calls, 1)
        e["hbm_read_MB_per_call_x2_corrected"] = round(rd / n / 1e6, 3)
        e["hbm_write_MB_per_call"] = round(wr / n / 1e6, 3)
        e["hbm_GBps_over_device_time"] = round((rd + wr) / (us * 1e-6) / 1e9, 1) if us > 0 else None
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
print(f"{'kernel':<70} {'calls':>6} {'us tot':>9} {'%':>6} {'mfma':>6} {'ldscf':>6} {'GB/s':>7}")
for k, e in list(res.items())[:28]:
    print(f"{k[:70]:<70} {e['calls']:>6} {e['device_us_total']:>9.0f} {e['pct_of_device_time']:>6.2f} "
          f"{e.get('mfma_busy_frac', float('nan')):>6.3f} {e.get('lds_bank_conflict_frac_of_lds_active', float('nan')):>6.3f} "
          f"{(e.get('hbm_GBps_over_device_time') or float('nan')):>7.0f}")
