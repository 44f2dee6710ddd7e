#!/usr/bin/env python3
"""Times pvol_preprocess (device photon shooter + search-structure build) on one of the golden scenes.

    python tools/time_shoot.py [scene] [n_photons] [n_tasks]
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import torch  # noqa: F401  (load torch's HIP runtime first, as tests/conftest.py does)
except Exception:
    pass


def main():
    scene_name = sys.argv[1] if len(sys.argv) > 1 else "volumescene_h"
    n_photons = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    n_tasks = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
    pkg = importlib.import_module("cs348b-pbrt_amd")
    pvol = importlib.import_module("cs348b-pbrt_amd.pvol")
    abi, blob = pkg.abi, pkg.blob
    s = blob.load(os.path.join(ROOT, "tests", "golden", "scene_%s.bin" % scene_name))
    p = abi.params_from_blob(s, n_volume_photons=n_photons)
    pv = pvol.PhotonVolume(p)
    pv.set_scene(abi.SceneHolder(s))
    t0 = time.perf_counter()
    pv.preprocess(n_tasks)
    wall = time.perf_counter() - t0
    shoot_s, build_s = pv.preprocess_times()
    st = pv.shoot_stats()
    print(json.dumps({"scene": scene_name, "requested": n_photons, "n_tasks": n_tasks, "stored": pv.photon_count(), "wall_s": wall,
                      "shoot_s": shoot_s, "grid_build_s": build_s, "stored_Mphotons_per_s": pv.photon_count() / shoot_s / 1e6,
                      "Mpaths_per_s": st["paths"] / shoot_s / 1e6, "Mmarch_steps_per_s": st["march_steps"] / shoot_s / 1e6,
                      "shoot_wpe": os.environ.get("PVOL_SHOOT_WPE", "2"), "stats": st}))
    pv.close()


if __name__ == "__main__":
    main()
