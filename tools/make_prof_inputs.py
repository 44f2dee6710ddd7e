#!/usr/bin/env python3
"""Writes the plain-file inputs of tools/pvol_prof (the native driver used under rocprofv3 --pmc):
the bench.py workload (volumescene-homogeneous frame, synthetic 1 M-photon map) at a given spp.

    python tools/make_prof_inputs.py OUTDIR [--spp 4] [--photons 1000000]
"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--photons", type=int, default=1000000)
    ap.add_argument("--xres", type=int, default=1280)
    ap.add_argument("--yres", type=int, default=720)
    ap.add_argument("--scene", default="volumescene_h", help="volumescene_h (the bench's C2) or pinkfloyd (C3: nused 500, two lights; camera rays unclipped)")
    ap.add_argument("--shoot-tasks", type=int, default=16384)
    a = ap.parse_args()
    import torch
    pkg = importlib.import_module("cs348b-pbrt_amd")
    abi, blob = pkg.abi, pkg.blob
    os.makedirs(a.out, exist_ok=True)
    scene = blob.load(os.path.join(bench.GOLD, "scene_%s.bin" % a.scene))
    holder = abi.SceneHolder(scene)
    params = abi.params_from_blob(scene, n_volume_photons=a.photons) if a.scene == "volumescene_h" else abi.params_from_blob(scene, n_volume_photons=a.photons, n_caustic_photons=0)
    with open(os.path.join(a.out, "scene.bin"), "wb") as f:
        f.write(bytes(holder.scene))
        f.write(bytes(holder.lights)[:C.sizeof(abi.Light) * holder.scene.n_lights])
        f.write(bytes(holder.tris)[:C.sizeof(abi.Triangle) * holder.scene.n_triangles])
        f.write(bytes(holder.mats)[:C.sizeof(abi.Material) * holder.scene.n_materials])
        if holder.density is not None:
            f.write(holder.density.tobytes())
    with open(os.path.join(a.out, "params.bin"), "wb") as f:
        f.write(bytes(params))
    # the bench's own map: shot on the device with the bench's task count (falls back to the resampled map without a GPU)
    try:
        pvol = importlib.import_module("cs348b-pbrt_amd.pvol")
        pv = pvol.PhotonVolume(params)
        pv.set_scene(holder)
        pv.preprocess(a.shoot_tasks)
        p, w, al = pv.download_photons()
        pv.close()
    except Exception as e:   # noqa: BLE001
        print("device shoot unavailable (%s): synthetic map" % e)
        p, w, al = bench.synth_photons(a.photons)
    with open(os.path.join(a.out, "photons.bin"), "wb") as f:
        f.write(np.uint32(len(p)).tobytes())
        f.write(p.tobytes()); f.write(w.tobytes()); f.write(al.tobytes())
    x0s, x1s, y0s, y1s, n_tiles = bench.frame_tiles(a.xres, a.yres)
    if a.scene == "volumescene_h":
        rays, counts = bench.build_rays(torch, torch.device("cpu"), scene, a.xres, a.yres, a.spp, (x0s, x1s, y0s, y1s), seed=1234)
        st = abi.make_streams(np.arange(n_tiles, dtype=np.uint32), counts.astype(np.uint32))
        r = rays.numpy()
    else:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import measure_configs
        r, st = measure_configs.rays_for(scene, a.xres, a.yres, a.spp, 7)
    with open(os.path.join(a.out, "rays.bin"), "wb") as f:
        f.write(np.uint32(len(r)).tobytes()); f.write(np.uint32(len(st)).tobytes())
        f.write(r.tobytes()); f.write(st.tobytes())
    print("wrote %s: %d rays, %d streams, %d photons" % (a.out, len(r), len(st), len(p)))


if __name__ == "__main__":
    main()
