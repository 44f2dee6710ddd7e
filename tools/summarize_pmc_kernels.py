#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc passes (tools/run_profiles_c3.sh) -> profiles/<tag>_pmc_summary.json.

    python tools/summarize_pmc_kernels.py r03_c3

For every kernel of the run: counter sums, instruction mix per wave, VALU pipe busy fraction, HBM bytes
((2 x FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE counts half the bytes on gfx950, MI355X_MICROARCH.md), L2 hit rate, and the time
from the kernel trace of the same command."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
g = os.path.join(ROOT, "gpurun_out")


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sorted(glob.glob(os.path.join(g, tag + "_pmc_*"))):
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not os.path.isdir(d) or not files:
        continue
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
times = {}
ks = glob.glob(os.path.join(g, tag + "_trace", "*", "*kernel_stats.csv"))
if ks:
    for r in csv.DictReader(open(max(ks, key=os.path.getmtime))):
        times[short(r["Name"])] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6, "percent": float(r["Percentage"])}
plain = None
pj = os.path.join(g, tag + "_prof_plain.json")
if os.path.exists(pj):
    for line in open(pj):
        if line.startswith("{"):
            plain = json.loads(line)
out = {"driver": "tools/pvol_prof via tools/run_profiles_c3.sh (pinkfloyd 1920x1080, nused 500, two lights; Li() alone over unclipped camera rays)", "run": plain, "kernels": {}}
CLK = 2.4e9
for k, c in sorted(agg.items(), key=lambda kv: -times.get(kv[0], {}).get("total_ms", 0.0)):
    if times.get(k, {}).get("percent", 0.0) < 0.5:
        continue
    t = times[k]["total_ms"] * 1e-3
    hbm = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
    waves = max(1.0, c.get("SQ_WAVES", 0.0))
    out["kernels"][k] = {
        "time": times[k], "counters": dict(c),
        "per_wave": {n: c.get(n, 0.0) / waves for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")},
        "valu_issue_frac_of_peak": c.get("SQ_INSTS_VALU", 0.0) / t / (1024 * CLK / 2) if t else None,
        "valu_pipe_busy": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024 * t * CLK) if t else None,
        "valu_cycles_per_inst": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / max(1.0, c.get("SQ_INSTS_VALU", 0.0)),
        "wait_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0.0)) if c.get("SQ_WAVE_CYCLES") else None,
        "lds_wait_frac": c.get("SQ_WAIT_INST_LDS", 0.0) / max(1.0, c.get("SQ_WAIT_INST_ANY", 0.0)) if c.get("SQ_WAIT_INST_ANY") else None,
        "hbm_bytes": hbm, "hbm_GBps": hbm / t / 1e9 if t else None, "hbm_frac_of_8TBps": hbm / t / 8e12 if t else None,
        "l2_hit_rate": c.get("TCC_HIT_sum", 0.0) / max(1.0, c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)),
    }
    m = out["kernels"][k]
    bound = "valu_issue" if (m["valu_pipe_busy"] or 0) > 0.5 else ("hbm" if (m["hbm_frac_of_8TBps"] or 0) > 0.4 else "latency (neither the VALU pipe nor HBM is near its roof)")
    m["roofline"] = {"bound": bound, "valu_pipe_busy": m["valu_pipe_busy"], "hbm_frac": m["hbm_frac_of_8TBps"]}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json"), "w"), indent=1)
if ks:
    rows = list(csv.reader(open(max(ks, key=os.path.getmtime))))
    keep = [rows[0]] + [r for r in rows[1:] if "rocprim" not in r[0] and not r[0].startswith("void at::")][:12]
    csv.writer(open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"), "w", newline="")).writerows(keep)
for k, m in out["kernels"].items():
    print("%-40s %8.1f ms  VALU busy %.2f  issue %.2f  HBM %.3f  L2 hit %.2f  wait %.2f" % (k[:40], m["time"]["total_ms"], m["valu_pipe_busy"] or 0, m["valu_issue_frac_of_peak"] or 0,
                                                                                 m["hbm_frac_of_8TBps"] or 0, m["l2_hit_rate"], m["wait_frac_of_wave_cycles"] or 0))
