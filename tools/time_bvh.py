#!/usr/bin/env python3
"""Row f4 measurement: device LBVH build time and what a frame costs when every closest / any hit walks the hierarchy.

    python tools/time_bvh.py [nu,nv ...]  > profiles/r02_bvh.jsonl      (defaults: 0 64,32 256,128 1024,512 2048,1024,b -- `,b`: build only)

Scene: the volumescene room (6 wall triangles) with a rippled ball of 2 nu nv - 2 nu triangles in the medium; `0` is the room
alone (linear scan out of the scalar cache) as the point of comparison.  Per size: pvol_set_scene wall time, the build's own
HIP-event time (pvol_get_accel_info), then whole render tasks (sampler + camera pre-pass with the closest-hit clip, Li with one
shadow ray per march step, film) on a 640x360 frame at 16 spp over a device-shot 100 k-photon map.
"""
import importlib
import json
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


def bumpy_ball(nu, nv, centre, radius):
    th = np.pi * np.arange(nv + 1) / nv
    ph = 2 * np.pi * np.arange(nu) / nu
    T, P = np.meshgrid(th, ph, indexing="ij")
    r = radius * (1 + 0.12 * np.sin(9 * T) * np.sin(7 * P))
    V = np.stack([centre[0] + r * np.sin(T) * np.cos(P), centre[1] + r * np.cos(T), centre[2] + r * np.sin(T) * np.sin(P)], -1)
    V = V.astype(np.float32).reshape(-1, 3)
    j, i = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a, b = j * nu + i, j * nu + (i + 1) % nu
    c, d = (j + 1) * nu + (i + 1) % nu, (j + 1) * nu + i
    idx = np.concatenate([np.stack([a, b, c], -1)[j > 0], np.stack([a, c, d], -1)[j < nv - 1]])
    return V[idx].reshape(-1, 9)


def scene_with_ball(nu, nv):
    s = dict(blob.load(os.path.join(bench.GOLD, "scene_meshroom.bin")))
    walls = s["tris.p"].reshape(-1, 9)[:6]
    ball = bumpy_ball(nu, nv, (0.5, 1.3, 4.0), 0.9) if nu else np.zeros((0, 9), np.float32)
    s["tris.p"] = np.concatenate([walls, ball]).astype(np.float32).reshape(-1)
    s["tris.material"] = np.concatenate([np.zeros(6, np.int32), np.ones(len(ball), np.int32)])
    s["tris.flip"] = np.zeros(6 + len(ball), np.int32)
    return s


def main():
    import torch
    sizes = [a for a in sys.argv[1:] if not a.startswith("-")] or ["0", "64,32", "256,128", "1024,512", "2048,1024,b"]
    xres, yres, spp = 640, 360, 16
    for sz in sizes:
        build_only = sz.endswith(",b")   # the largest size: set_scene only
        nu, nv = ([int(v) for v in sz.replace(",b", "").split(",")] + [0])[:2]
        s = scene_with_ball(nu, nv)
        n_tris = len(s["tris.material"])
        params = abi.params_from_blob(s, n_volume_photons=100000)
        pv = pvol.PhotonVolume(params)
        holder = abi.SceneHolder(s)
        t = time.perf_counter()
        pv.set_scene(holder)
        set_scene_s = time.perf_counter() - t
        n_bvh, build_ms = pv.accel_info()
        if build_only:
            print(json.dumps({"triangles": n_tris, "triangles_in_hierarchy": n_bvh, "lbvh_build_ms": build_ms,
                              "lbvh_build_Mtris_per_s": n_bvh / build_ms / 1e3, "set_scene_wall_s": set_scene_s}), flush=True)
            pv.close()
            continue
        t = time.perf_counter()
        pv.preprocess(2048)
        shoot_wall = time.perf_counter() - t
        shoot_s, grid_s = pv.preprocess_times()
        st = pv.shoot_stats()
        n_tiles = bench.frame_tiles(xres, yres)[4]
        cam = abi.perspective_camera(float(s["camera.fov"][0]), xres, yres, s["camera.c2w"])
        film = abi.make_film(xres, yres, pvol.gaussian_filter_table())
        smp = abi.make_sampler(xres, yres, spp, n_tiles)
        ids = np.arange(n_tiles, dtype=np.uint32)
        n = pvol.render_sample_count(smp, ids)
        px = torch.zeros((yres, xres, 4), dtype=torch.float32, device="cuda:0")
        frames = []
        for rep in range(2):   # the second frame is the measurement (work buffers allocated, code resident)
            px.zero_()
            torch.cuda.synchronize()
            pv.kernel_time_ms(reset=True)
            t = time.perf_counter()
            pv.render_tasks(cam, film, smp, ids, px.data_ptr())
            torch.cuda.synchronize()
            frames.append(time.perf_counter() - t)
        pv.check_errors()
        kms, nl = pv.kernel_time_ms()
        print(json.dumps({"triangles": n_tris, "triangles_in_hierarchy": n_bvh, "lbvh_build_ms": build_ms,
                          "lbvh_build_Mtris_per_s": (n_bvh / build_ms / 1e3) if build_ms > 0 else None, "set_scene_wall_s": set_scene_s,
                          "photons": pv.photon_count(), "shoot_s": shoot_s, "shoot_Mpaths_per_s": st["paths"] / shoot_s / 1e6,
                          "frame": {"xres": xres, "yres": yres, "spp": spp, "samples": int(n), "frame_s": frames[1],
                                    "Msamples_per_s_whole_pipeline": n / frames[1] / 1e6, "march_kernel": pv.march_kernel_name(),
                                    "march_kernels_s": kms * nl * 1e-3}}), flush=True)
        pv.close()


if __name__ == "__main__":
    main()
