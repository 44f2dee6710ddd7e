#!/usr/bin/env python3
"""Times the BASELINE.json configs other than the bench line on one MI355X (reduced sample counts where a
full run would take minutes; the reduction is recorded).  Output: one JSON object per config.

    python tools/measure_configs.py > profiles/r01_configs.jsonl
"""
import importlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module("cs348b-pbrt_amd")
pvol = importlib.import_module("cs348b-pbrt_amd.pvol")
abi, blob = pkg.abi, pkg.blob
GOLD = bench.GOLD


def rays_for(scene, xres, yres, spp, seed):
    """Jittered pinhole samples tile by tile (reference tile order), unclipped (maxt = inf)."""
    x0s, x1s, y0s, y1s, n_tiles = bench.frame_tiles(xres, yres)
    rng = np.random.default_rng(seed)
    c2w = scene["camera.c2w"].reshape(4, 4).astype(np.float64)
    tan = math.tan(math.radians(float(scene["camera.fov"][0])) / 2)
    aspect = xres / float(yres)
    sx, sy = (aspect, 1.0) if aspect > 1 else (1.0, 1.0 / aspect)
    chunks, counts = [], []
    for t in range(n_tiles):
        w, h = int(x1s[t] - x0s[t]), int(y1s[t] - y0s[t])
        n = max(0, w * h * spp)
        counts.append(n)
        if n == 0:
            continue
        gy, gx = np.meshgrid(np.arange(y0s[t], y1s[t]), np.arange(x0s[t], x1s[t]), indexing="ij")
        X = np.repeat(gx.reshape(-1), spp) + rng.random(n)
        Y = np.repeat(gy.reshape(-1), spp) + rng.random(n)
        d = np.stack([(2 * X / xres - 1) * sx * tan, (1 - 2 * Y / yres) * sy * tan, np.ones(n)], 1)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        chunks.append(d @ c2w[:3, :3].T)
    d = np.concatenate(chunks).astype(np.float32)
    r = abi.make_rays(np.zeros((len(d), 3), np.float32), d, 0.0, np.inf, rng.random(len(d)).astype(np.float32))
    st = abi.make_streams(np.arange(n_tiles, dtype=np.uint32), np.array(counts, np.uint32))
    return r, st


def run(name, scene_name, xres, yres, spp, n_photons, tasks, note, **over):
    scene = blob.load(os.path.join(GOLD, "scene_%s.bin" % scene_name))
    if "density_n" in over:
        n = over.pop("density_n")
        scene = dict(scene)
        scene["vol.dims"] = np.array([n, n, n], np.int32)
        g = (np.arange(n) + .5) / n
        zz, yy, xx = np.meshgrid(g, g, g, indexing="ij")
        rng = np.random.default_rng(348)
        dens = 0.5 + 0.5 * np.sin(7 * xx) * np.sin(5 * yy) * np.sin(3 * zz) + 0.25 * (rng.random((n, n, n)) - .5)
        scene["vol.density"] = np.clip(dens, 0, 1.5).astype(np.float32).reshape(-1)
    params = abi.params_from_blob(scene, n_volume_photons=n_photons, **over)
    pv = pvol.PhotonVolume(params)
    pv.set_scene(abi.SceneHolder(scene))
    t = time.perf_counter()
    pv.preprocess(tasks)
    t_shoot = time.perf_counter() - t
    st_sh = pv.shoot_stats()
    rec = {"config": name, "scene": scene_name, "note": note, "photons": pv.photon_count(), "shoot_tasks": tasks, "shoot_s": t_shoot,
           "shoot_Mpaths_per_s": st_sh["paths"] / t_shoot / 1e6, "shoot_Mphotons_per_s": st_sh["stored_volume"] / t_shoot / 1e6,
           "shoot_Mmarch_steps_per_s": st_sh["march_steps"] / t_shoot / 1e6}
    if spp > 0:
        rays, streams = rays_for(scene, xres, yres, spp, 7)
        pv.li(rays[:4096], abi.make_streams(np.array([0], np.uint32), np.array([4096], np.uint32)), abi.OUT_XYZ)   # warm up
        pv.kernel_time_ms(reset=True)
        t = time.perf_counter()
        out, draws = pv.li(rays, streams, abi.OUT_XYZ)
        wall = time.perf_counter() - t
        kms, _ = pv.kernel_time_ms()
        rec.update({"xres": xres, "yres": yres, "spp": spp, "li_calls": len(rays), "kernel_ms": kms, "Msamples_per_s_kernel": len(rays) / kms / 1e3,
                    "Msamples_per_s_host_api_incl_pcie": len(rays) / wall / 1e6, "mean_draws": float(draws.mean())})
    pv.close()
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    run("C1-h", "volumescene_h", 256, 256, 16, 100000, 4096, "config 1 with Volume homogeneous (gather on)")
    run("C1-literal", "volumescene_rainbow", 256, 256, 16, 100000, 4096, "config 1 as shipped (rainbow volume: no gather)")
    run("C3", "pinkfloyd", 480, 270, 4, 4000000, 1024, "config 3 at 1/16 resolution, 4 spp instead of 512, 1 GPU; two lights => resolve/replay kernels",
        n_caustic_photons=0)
    run("C4", "volumescene_grid16", 128, 128, 8, 2000000, 16384, "config 4: 128^3 VolumeGrid regenerated here, 128x128 at 8 spp instead of 256^2 at 1024",
        density_n=128)
    run("C5", "shootbench", 0, 0, 0, 100000000, 32768, "config 5: shoot only, 100 M photons requested")
