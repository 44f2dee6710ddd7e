#!/usr/bin/env python3
"""Which HIP / HSA / rocprofiler-register images does this process hold?  (Round-1 `rocprofv3 --pmc -- python3 bench.py`
runs died inside torch's own kernels; this lists the runtime copies mapped with and without the profiler in front.)

    python3 tools/show_runtime_maps.py
    rocprofv3 --kernel-trace -d /tmp/x -- python3 tools/show_runtime_maps.py
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def runtime_images():
    seen = []
    for line in open("/proc/self/maps"):
        path = line.split()[-1]
        if "r-xp" in line and any(k in path for k in ("amdhip64", "hsa-runtime", "rocprofiler", "libpvol")):
            if path not in seen:
                seen.append(path)
    return seen


before = runtime_images()
import torch  # noqa: E402

after_torch = runtime_images()
ok = torch.cuda.is_available()
pvol = importlib.import_module("cs348b-pbrt_amd.pvol")
n = pvol.lib().pvol_device_count()
print(json.dumps({"preloaded_by_the_launcher": before, "after_import_torch": after_torch, "after_loading_libpvol": runtime_images(),
                  "torch_sees_gpu": bool(ok), "libpvol_device_count": int(n),
                  "LD_PRELOAD": os.environ.get("LD_PRELOAD", ""), "ROCP_TOOL_LIBRARIES": os.environ.get("ROCP_TOOL_LIBRARIES", "")}, indent=1))
