#!/usr/bin/env python3
"""Condenses rocprofv3 outputs under gpurun_out/ into the small summaries committed under profiles/.

    python tools/summarize_pmc.py r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
g = os.path.join(ROOT, "gpurun_out")

agg = collections.defaultdict(float)
kernels = set()
for d in sorted(glob.glob(os.path.join(g, tag + "_pmc_*"))):
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not os.path.isdir(d) or not files:
        continue
    f = max(files, key=os.path.getmtime)   # gpurun merges runs: keep the newest pass only
    for r in csv.DictReader(open(f)):
        if "li_group_kernel" in r["Kernel_Name"]:   # the dominant kernel only (li_fixup_kernel's share is in the kernel trace)
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            kernels.add(r["Kernel_Name"].split("<")[0].replace("void ", ""))
plain = None
for f in glob.glob(os.path.join(g, tag + "_pmc_*.log")) + glob.glob(os.path.join(g, tag + "_prof_plain.json")):
    for line in open(f, errors="ignore"):
        if line.startswith("{") and "rays" in line:
            plain = json.loads(line)
if agg and plain:
    rays = plain["rays"]
    fetch_kb, write_kb = agg.get("FETCH_SIZE", 0.0), agg.get("WRITE_SIZE", 0.0)
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0   # FETCH_SIZE counts half the bytes on gfx950 (MI355X_MICROARCH.md, HBM)
    summary = {
        "kernel": "+".join(sorted(kernels)), "driver": "tools/pvol_prof (tools/run_profiles.sh: bench.py scene and photon-map recipe, 640x360 at 256 spp)", "rays": rays,
        "counters": dict(agg),
        "per_ray": {k: v / rays for k, v in agg.items()},
        "l2_hit_rate": agg.get("TCC_HIT_sum", 0) / max(1.0, agg.get("TCC_HIT_sum", 0) + agg.get("TCC_MISS_sum", 0)),
        "valu_cycles_per_inst": 4.0 * agg.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, agg.get("SQ_INSTS_VALU", 0)),
        "hbm_bytes": hbm, "hbm_bytes_per_ray": hbm / rays,
        "kernel_avg_ms_under_pmc": plain["kernel_avg_ms"],
    }
    json.dump(summary, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1)
    json.dump({"kernel": "li_group_kernel", "hbm_bytes_per_ray": hbm / rays, "valu_insts_per_ray": agg.get("SQ_INSTS_VALU", 0) / rays,
               "salu_insts_per_ray": agg.get("SQ_INSTS_SALU", 0) / rays,
               "valu_active_quadcycles_per_ray": agg.get("SQ_ACTIVE_INST_VALU", 0) / rays,
               "mfma_busy_cycles_per_ray": agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / rays, "mfma_insts_per_ray": agg.get("SQ_INSTS_MFMA", 0) / rays,
               "waves_per_simd": 3, "vgprs": 168,
               "source": "profiles/%s_pmc_summary.json: SQ_ACTIVE_INST_VALU / rays, SQ_INSTS_VALU / rays and (2*FETCH_SIZE + WRITE_SIZE)*1024 / rays, rocprofv3 --pmc passes of tools/pvol_prof on the bench's scene and device-shot map at 640x360, 256 spp" % tag},
              open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("hbm bytes/ray %.0f, L2 hit %.3f, VALU insts/ray %.0f" % (hbm / rays, summary["l2_hit_rate"], agg.get("SQ_INSTS_VALU", 0) / rays))
_ks = glob.glob(os.path.join(g, tag + "_trace", "*", "*kernel_stats.csv"))
for f in ([max(_ks, key=os.path.getmtime)] if _ks else []):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if not r[0].startswith("void at::") and "rocprim" not in r[0] and "anonymous" not in r[0]][:12]
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as fo:
        csv.writer(fo).writerows(keep)
    print("kernel stats:", keep[1][:4])
for name in glob.glob(os.path.join(g, tag + "_bench_*.json")):
    lines = [l for l in open(name) if l.startswith("{")]
    if lines:
        open(os.path.join(out, os.path.basename(name)), "w").write(lines[-1])
