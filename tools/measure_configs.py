#!/usr/bin/env python3
"""Times the BASELINE.json configs other than the bench line on one MI355X.  Output: one JSON object per config.

    python tools/measure_configs.py [C1 C3 C4 C5 ...] [--full]  > profiles/r02_configs.jsonl

Two measurements per render config:
  * `li`     Li() alone through the host API (pvol_li_batch) on a reduced frame: the march + gather kernels (HIP-event time)
  * `frame`  whole SamplerRendererTasks on the device (pvol_render_tasks_device: sampler + camera pre-pass, Li, film) at the
             size the line states -- with --full the config's own size, otherwise a reduced sample count (recorded)
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


def grid_scene(scene, n):
    """density(x,y,z) = clamp(0.5 + 0.5 sin(7x) sin(5y) sin(3z) + 0.25 noise(seed 348), 0, 1.5) on an n^3 lattice."""
    scene = dict(scene)
    scene["vol.dims"] = np.array([n, n, n], np.int32)
    g = (np.arange(n) + .5) / n
    zz, yy, xx = np.meshgrid(g, g, g, indexing="ij")
    rng = np.random.default_rng(348)
    dens = 0.5 + 0.5 * np.sin(7 * xx) * np.sin(5 * yy) * np.sin(3 * zz) + 0.25 * (rng.random((n, n, n)) - .5)
    scene["vol.density"] = np.clip(dens, 0, 1.5).astype(np.float32).reshape(-1)
    return scene


def parity_leg(torch, pv, scene, params, xres, yres, spp, n_tiles, n_pick=16, spp_cap=16):
    """In-run parity of a frame line: whole render tasks of THIS frame (same photon map -- downloaded from the device --,
    same parameters, same task windows and seeds) through the oracle's SamplerRendererTask loop and through the device tile driver, at
    min(spp, spp_cap) samples per pixel so that the oracle finishes in a minute.  Per-pixel relative L2 (the north_star's bar: 1e-4) and
    the RNG stream end of every task (exact)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    spp_p = min(spp, spp_cap)
    cam = abi.perspective_camera(float(scene["camera.fov"][0]), xres, yres, scene["camera.c2w"])
    film = abi.make_film(xres, yres, pvol.gaussian_filter_table())
    smp = abi.make_sampler(xres, yres, spp_p, n_tiles)
    pick = np.unique(np.linspace(0, n_tiles - 1, n_pick).astype(np.uint32))
    cores = max(1, min(os.cpu_count() or 1, 16))
    o = orc.Oracle(abi.SceneHolder(scene), params)
    o.set_photons(*pv.download_photons())
    t0 = time.perf_counter()
    ref = orc.render_tasks(o, cam, film, smp, pick, records=True, n_threads=cores)
    t_cpu = time.perf_counter() - t0
    n = len(ref["xyzT"])
    dev = torch.device("cuda:0")
    px = torch.zeros((yres, xres, 4), dtype=torch.float32, device=dev)
    xyz = torch.zeros((max(n, 1), 4), dtype=torch.float32, device=dev)
    streams = torch.zeros((len(pick), 32), dtype=torch.uint8, device=dev)
    pv.render_tasks(cam, film, smp, pick, px.data_ptr(), abi.RenderDebug(0, 0, xyz.data_ptr(), streams.data_ptr()))
    torch.cuda.synchronize()
    pv.check_errors()
    got = xyz.cpu().numpy().astype(np.float64)[:n]
    r = ref["xyzT"].astype(np.float64)
    end = streams.cpu().numpy().view(abi.STREAM_DTYPE).reshape(-1)["end_draw"]
    scale = max(float(np.abs(r[:, :3]).max()), 1e-30)
    err = np.linalg.norm(got[:, :3] - r[:, :3], axis=1) / np.maximum(np.linalg.norm(r[:, :3], axis=1), 1e-6 * scale)
    gp, rp = got[:, :3].reshape(-1, spp_p, 3).mean(1), r[:, :3].reshape(-1, spp_p, 3).mean(1)
    perr = np.linalg.norm(gp - rp, axis=1) / np.maximum(np.linalg.norm(rp, axis=1), 1e-6 * scale)
    return {"tasks": int(len(pick)), "samples": int(n), "spp": int(spp_p), "oracle_s": t_cpu, "oracle_threads": cores,
            "max_rel_l2_per_pixel": float(perr.max()), "max_rel_l2_per_sample": float(err.max()), "samples_above_1e-4": int((err > 1e-4).sum()),
            "rng_end_positions_equal": bool((end == ref["end_draws"]).all()), "tolerance_per_pixel": 1e-4,
            "ok": bool(perr.max() <= 1e-4 and (end == ref["end_draws"]).all())}


def run(name, scene_name, n_photons, tasks, note, li=None, frame=None, density_n=0, block_paths=4096, **over):
    import torch
    scene = blob.load(os.path.join(GOLD, "scene_%s.bin" % scene_name))
    if density_n:
        scene = grid_scene(scene, density_n)
    params = abi.params_from_blob(scene, n_volume_photons=n_photons, **over)
    pv = pvol.PhotonVolume(params)
    pv.set_scene(abi.SceneHolder(scene))
    t = time.perf_counter()
    pv.preprocess(tasks, block_paths)
    t_shoot = time.perf_counter() - t
    shoot_s, build_s = pv.preprocess_times()
    st_sh = pv.shoot_stats()
    rec = {"config": name, "scene": scene_name, "note": note, "photons": pv.photon_count(), "photons_requested": n_photons, "shoot_tasks": tasks, "shoot_block_paths": block_paths,
           "shoot_s": shoot_s, "grid_build_s": build_s, "shoot_Mpaths_per_s": st_sh["paths"] / shoot_s / 1e6,
           "shoot_stored_Mphotons_per_s": st_sh["stored_volume"] / shoot_s / 1e6, "shoot_Mmarch_steps_per_s": st_sh["march_steps"] / shoot_s / 1e6,
           "nused": params.n_used, "maxdist": params.max_dist, "stepsize": params.step_size}
    if li:
        xres, yres, spp = li
        rays, streams = rays_for(scene, xres, yres, spp, 7)
        pv.li(rays[:4096], abi.make_streams(np.array([0], np.uint32), np.array([4096], np.uint32)), abi.OUT_XYZ)   # warm up
        pv.kernel_time_ms(reset=True)
        pv.stats(reset=True)
        t = time.perf_counter()
        out, draws = pv.li(rays, streams, abi.OUT_XYZ)
        wall = time.perf_counter() - t
        kms, _ = pv.kernel_time_ms()
        work = pv.stats()
        rec["li"] = {"xres": xres, "yres": yres, "spp": spp, "li_calls": len(rays), "kernel": pv.march_kernel_name(), "kernel_ms": kms,
                     "Msamples_per_s_kernel": len(rays) / kms / 1e3, "Msamples_per_s_host_api_incl_pcie": len(rays) / wall / 1e6,
                     "mean_draws": float(draws.mean()), "steps_per_ray": work["n_steps"] / max(1, work["n_rays"])}
    if frame:
        xres, yres, spp, stated = frame
        n_tiles = bench.frame_tiles(xres, yres)[4]
        cam = abi.perspective_camera(float(scene["camera.fov"][0]), xres, yres, scene["camera.c2w"])
        film = abi.make_film(xres, yres, pvol.gaussian_filter_table())
        smp = abi.make_sampler(xres, yres, spp, n_tiles)
        ids = np.arange(n_tiles, dtype=np.uint32)
        n = pvol.render_sample_count(smp, ids)
        dev = torch.device("cuda:0")
        px = torch.zeros((yres, xres, 4), dtype=torch.float32, device=dev)
        rgb = torch.zeros((yres, xres, 3), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        pv.kernel_time_ms(reset=True)
        pv.stats(reset=True)
        t = time.perf_counter()
        pv.render_tasks(cam, film, smp, ids, px.data_ptr())
        pv.film_resolve(film, px.data_ptr(), rgb.data_ptr())
        torch.cuda.synchronize()
        wall = time.perf_counter() - t
        pv.check_errors()
        kms, nl = pv.kernel_time_ms()
        work = pv.stats()
        rec["frame"] = {"xres": xres, "yres": yres, "spp": spp, "stated_size": stated, "render_tasks": int(n_tiles), "samples": int(n),
                        "kernel": pv.march_kernel_name(), "frame_s": wall, "Msamples_per_s_whole_pipeline": n / wall / 1e6,
                        "march_kernels_s": kms * nl * 1e-3, "Msamples_per_s_march_kernels": n / (kms * nl) / 1e3,
                        "mean_rgb": float(rgb.mean().item()),
                        # nused beyond the bucket plan (li_fixup_group_kernel): lookups served from shared buckets / by the exact pass
                        "handed_over_shared_bucket": work["group_plan_skipped"], "handed_over_exact_pass": work["group_guess_failed"],
                        "bucket_stagings": work["cy_fallback"],
                        "exact_pass_because": { "counter_wrapped": work["group_deferred_overflow"],
                                               "crowded_sub_bin": work["group_deferred_too_few"], "radius_corrections_exhausted": work["group_attempts"]}}
        if "--raw-stats" in sys.argv:
            rec["frame"]["raw_stats"] = work
        if "--no-parity" not in sys.argv:
            rec["frame"]["parity"] = parity_leg(torch, pv, scene, params, xres, yres, spp, n_tiles)
    pv.close()
    print(json.dumps(rec), flush=True)


def heartbeat(period=45.0):
    """A line on stderr every `period` seconds: full-size frames run for minutes without output of their own."""
    import threading
    t0 = time.time()

    def beat():
        while True:
            time.sleep(period)
            print("[measure_configs] %.0f s" % (time.time() - t0), file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()


if __name__ == "__main__":
    heartbeat()
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    full = "--full" in sys.argv
    spp_over = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--spp=")]   # reduced frames at another sample count
    if "--no-li" in sys.argv:   # frames only (kernel traces)
        _run = run
        def run(*a, **k):   # noqa: E731,F811
            k.pop("li", None)
            return _run(*a, **k)
    want = lambda c: not args or c in args   # noqa: E731
    if want("C1"):
        run("C1-h", "volumescene_h", 100000, 2048, "config 1 with Volume homogeneous (gather on), stated size", li=(256, 256, 16), frame=(256, 256, 16, True))
        run("C1-literal", "volumescene_rainbow", 100000, 2048, "config 1 as shipped (rainbow volume: no gather), stated size", li=(256, 256, 16),
            frame=(256, 256, 16, True))
    if want("C3"):
        # map: 4096 virtual tasks x 32-path blocks (pvol_preprocess_blocks): 4096 waves, a round is 131 k paths (round 2: 256 tasks x 4096)
        run("C3", "pinkfloyd", 4000000, 4096, "config 3: pinkfloyd 1920x1080, 4 M photons, nused 500, two lights, on ONE MI355X; frame at %s" %
            ("512 spp (stated)" if full else "%d spp instead of 512" % (spp_over[0] if spp_over else 8)), li=(480, 270, 4),
            frame=(1920, 1080, 512 if full else (spp_over[0] if spp_over else 8), full), n_caustic_photons=0, block_paths=32)
    if want("C4"):
        run("C4", "volumescene_grid16", 2000000, 16384, "config 4: 128^3 VolumeGrid (generated here), 2 M photons; frame 256x256 at %s" %
            ("1024 spp (stated)" if full else "64 spp instead of 1024"), li=(128, 128, 8),
            frame=(256, 256, 1024 if full else 64, full), density_n=128)
    if want("C5"):
        run("C5", "shootbench", 100000000, 2048, "config 5: shoot only, 100 M stored photons requested, device grid build timed separately")
