#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: Msamples/s of the volumetric photon-mapping hot path
(ray march + k-NN photon gather, PhotonVolumeIntegrator::Li) on config[1]:
projectScene/volumescene (homogeneous variant) at 1280x720, 256 spp, 1 M volume photons.

One "step" = one pass of the hot path over the whole frame's camera samples (incl. the filter apron:
1285 x 725 x 256 = 238.5 M Li() calls), everything resident in HBM: the default `--driver tile` runs whole
SamplerRendererTasks on the device (LD sampler + camera pre-pass, Li, film), `--driver batch` times Li() alone
over pre-built rays.  N > 1: the reference's render tasks (one MT19937 stream each) are sharded over ranks,
photon map replicated; weak scaling (default) has no data-path collective, `--strong` all-reduces the film over
RCCL; value = samples of all ranks / max-over-ranks time.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def tile_grid(n_tiles, dx, dy):
    """Sampler::ComputeSubWindow's tile lattice (core/sampler.cpp:55-74)."""
    nx, ny = n_tiles, 1
    while (nx & 1) == 0 and 2 * dx * ny < dy * nx:
        nx >>= 1
        ny <<= 1
    return nx, ny


def round_up_pow2(v):
    return 1 << (int(v) - 1).bit_length()


def frame_tiles(xres, yres):
    """The reference's render tiles over the sample extent (film/image.cpp:157-166: pixels +- filter
    width 2; renderers/samplerrenderer.cpp:206-208 with <= 128 cores; core/sampler.cpp:55-74)."""
    # GetSampleExtent: Floor2Int(0.5 - 2) = -2 ... Ceil2Int(0.5 + res + 2) = res + 3
    x_lo, x_hi, y_lo, y_hi = -2, xres + 3, -2, yres + 3
    n_tiles = round_up_pow2(max(32 * 1, xres * yres // 256))
    nx, ny = tile_grid(n_tiles, x_hi - x_lo, y_hi - y_lo)
    tids = np.arange(n_tiles)
    tx, ty = tids % nx, tids // nx
    f32 = np.float32

    def lerp_floor(k, n, lo, hi):   # Floor2Int(Lerp(float(k)/float(n), lo, hi)) in fp32, sampler.cpp:66-73
        t = k.astype(f32) / f32(n)
        return np.floor((f32(1) - t) * f32(lo) + t * f32(hi)).astype(np.int64)
    return (lerp_floor(tx, nx, x_lo, x_hi), lerp_floor(tx + 1, nx, x_lo, x_hi), lerp_floor(ty, ny, y_lo, y_hi),
            lerp_floor(ty + 1, ny, y_lo, y_hi), n_tiles)


def synth_photons(n, seed=348):
    """Synthetic 1 M-photon map shaped like a real one: the committed 6 k-photon map of this scene
    (shot by the oracle shooter) resampled with a 0.25-unit Gaussian jitter, flux rescaled."""
    pkg = importlib.import_module("cs348b-pbrt_amd")
    b = pkg.blob.load(os.path.join(GOLD, "photons_vh.bin"))
    P, W, A = b["p"].reshape(-1, 3), b["wi"].reshape(-1, 3), b["alpha"].reshape(-1, 30)
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, len(P), n)
    p = (P[idx] + rng.normal(0, 0.25, (n, 3))).astype(np.float32)
    lo, hi = np.array([-5, -0.5, -1.5], np.float32), np.array([5, 4.5, 6.5], np.float32)
    p = np.clip(p, lo + 1e-3, hi - 1e-3)
    w = W[rng.integers(0, len(P), n)].astype(np.float32)
    a = (A[idx] * (len(P) / float(n))).astype(np.float32)
    return p, w, a


def build_rays(torch, dev, scene, xres, yres, spp, tiles, seed):
    """Camera samples for tiles [t0, t1) in the reference's order (tile by tile, pixel by pixel, spp
    consecutive samples per pixel), as pvol_ray records on the device.  Synthetic jittered samples;
    rays are clipped at the scene's three quads like SamplerRenderer::Li does before the volume
    integrator runs (renderers/samplerrenderer.cpp:234)."""
    x0s, x1s, y0s, y1s = tiles
    counts = ((x1s - x0s) * (y1s - y0s) * spp).astype(np.int64)
    total = int(counts.sum())
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rays = torch.zeros((total, 12), dtype=torch.float32, device=dev)
    # pixel coordinates per sample, built tile by tile on the device in chunks
    off = 0
    tan = math.tan(math.radians(float(scene["camera.fov"][0])) / 2)
    aspect = xres / float(yres)
    sx, sy = (aspect, 1.0) if aspect > 1 else (1.0, 1.0 / aspect)
    chunk = 256
    for c0 in range(0, len(x0s), chunk):
        xs, ys = [], []
        for t in range(c0, min(c0 + chunk, len(x0s))):
            w, h = int(x1s[t] - x0s[t]), int(y1s[t] - y0s[t])
            if w <= 0 or h <= 0:
                continue
            px = torch.arange(int(x0s[t]), int(x1s[t]), device=dev, dtype=torch.float32)
            py = torch.arange(int(y0s[t]), int(y1s[t]), device=dev, dtype=torch.float32)
            gy, gx = torch.meshgrid(py, px, indexing="ij")
            xs.append(gx.reshape(-1).repeat_interleave(spp))
            ys.append(gy.reshape(-1).repeat_interleave(spp))
        if not xs:
            continue
        X = torch.cat(xs)
        Y = torch.cat(ys)
        n = X.numel()
        X = X + torch.rand(n, device=dev, generator=g)
        Y = Y + torch.rand(n, device=dev, generator=g)
        # perspective camera at the origin looking down +z (cameras/perspective.cpp), fov on the short axis
        dxn = (2 * X / xres - 1) * sx * tan
        dyn = (1 - 2 * Y / yres) * sy * tan
        d = torch.stack([dxn, dyn, torch.ones_like(dxn)], 1)
        d = d / d.norm(dim=1, keepdim=True)
        # closest hit with the three quads of volumescene (world space: floor y=-0.5, back z=6.5, side x=5)
        inf = torch.full((n,), float("inf"), device=dev)
        t_floor = torch.where(d[:, 1] < 0, -0.5 / d[:, 1], inf)
        hx, hz = d[:, 0] * t_floor, d[:, 2] * t_floor
        t_floor = torch.where((hx.abs() <= 5) & (hz >= -1.5) & (hz <= 8.5), t_floor, inf)
        t_back = 6.5 / d[:, 2]
        bx, by = d[:, 0] * t_back, d[:, 1] * t_back
        t_back = torch.where((bx.abs() <= 5) & (by >= -0.5) & (by <= 9.5), t_back, inf)
        t_side = torch.where(d[:, 0] > 0, 5.0 / d[:, 0], inf)
        sy_, sz_ = d[:, 1] * t_side, d[:, 2] * t_side
        t_side = torch.where((sy_ >= -0.5) & (sy_ <= 9.5) & (sz_ >= 0.5) & (sz_ <= 6.5), t_side, inf)
        maxt = torch.minimum(torch.minimum(t_floor, t_back), t_side)
        r = rays[off:off + n]
        r[:, 4:7] = d
        r[:, 7] = maxt
        r[:, 9] = torch.rand(n, device=dev, generator=g)   # scatter_u
        off += n
    assert off == total
    return rays, counts


def cpu_baseline(scene, params, photons, xres, yres, budget_s=15.0):
    """The oracle (CPU restatement of the reference path, kd-tree gather) timed on this host's cores
    on a bounded sample of the SAME workload: whole render tiles of the frame, one thread per core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    pkg = importlib.import_module("cs348b-pbrt_amd")
    abi = pkg.abi
    cores = max(1, min(os.cpu_count() or 1, 32))
    o = orc.Oracle(abi.SceneHolder(scene), params)
    t0 = time.time()
    o.set_photons(*photons)
    t_build = time.time() - t0
    # whole render tiles of the same frame, picked at random, same ray generator (incl. the surface clip) as the GPU pass
    import torch
    x0s, x1s, y0s, y1s, n_tiles = frame_tiles(xres, yres)
    rng = np.random.default_rng(1)
    n_streams = cores * 2
    spp = 1
    done, elapsed = 0, 0.0
    while elapsed < budget_s:
        pick = rng.choice(n_tiles, n_streams, replace=False)
        rays_t, counts = build_rays(torch, torch.device("cpu"), scene, xres, yres, spp, (x0s[pick], x1s[pick], y0s[pick], y1s[pick]),
                                    seed=int(rng.integers(1 << 30)))
        rn = rays_t.numpy()
        rays = np.zeros(len(rn), abi.RAY_DTYPE)
        rays["o"], rays["mint"], rays["d"], rays["maxt"], rays["time"], rays["scatter_u"] = rn[:, 0:3], rn[:, 3], rn[:, 4:7], rn[:, 7], rn[:, 8], rn[:, 9]
        st = abi.make_streams(pick.astype(np.uint32), counts.astype(np.uint32))
        t0 = time.time()
        o.li_batch(rays, st, abi.OUT_XYZ, n_threads=cores)
        dt = time.time() - t0
        done += len(rays)
        elapsed += dt
        if dt < 2.0:
            spp *= 2
    ctr = o.counters()
    return {"value": done / elapsed / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d Li() calls (random whole render tiles of the same frame, same ray generator/scene/photon map/params; kd-tree gather as core/kdtree.h), "
                      "%.1f s; kd build %.1f s; V=%.0f nodes, K=%.1f photons per lookup" %
                      (done, elapsed, t_build, ctr["n_nodes_visited"] / max(1, ctr["n_lookups"]), ctr["n_kept"] / max(1, ctr["n_lookups"]))}, ctr


def cpu_baseline_render(scene, params, photons, cam, film, smp, budget_s=20.0):
    """CPU baseline of the SAME pipeline (tile driver): the oracle's SamplerRendererTask loop (LD sampler, camera,
    Li with the kd-tree gather, film) on random whole render tasks of the same frame, all host cores, ~20 s."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    pkg = importlib.import_module("cs348b-pbrt_amd")
    abi = pkg.abi
    cores = max(1, min(os.cpu_count() or 1, 16))   # the GPU box grants a 16-core CPU share per GPU
    o = orc.Oracle(abi.SceneHolder(scene), params)
    t0 = time.time()
    o.set_photons(*photons)
    t_build = time.time() - t0
    o.counters(reset=True)
    rng = np.random.default_rng(7)
    n_tasks = smp.n_tasks
    # bounded sample: the oracle runs one render task per thread (a task is sequential: one RNG stream), so a batch needs at
    # least `cores` tasks to use the cores it reports; small tasks are batched further up to ~0.25 M samples
    per_task = (film.x_resolution + 5) * (film.y_resolution + 5) * smp.pixel_samples / n_tasks
    batch = int(max(1, min(n_tasks, max(cores, round(250000.0 / per_task)))))
    done, elapsed = 0, 0.0
    while elapsed < budget_s:
        pick = rng.choice(n_tasks, min(batch, n_tasks), replace=False).astype(np.uint32)
        t0 = time.time()
        r = orc.render_tasks(o, cam, film, smp, pick, records=False, n_threads=cores)
        dt = time.time() - t0
        done += r["n_samples"]
        elapsed += dt
        if dt < 3.0:
            batch *= 2
    ctr = o.counters()
    return {"value": done / elapsed / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d camera samples = random whole render tasks of the same frame through the oracle's SamplerRendererTask loop "
                      "(LDSampler, camera, Li with the kd-tree gather of core/kdtree.h, ImageFilm; same scene/photon map/params), "
                      "%.1f s on %d threads; kd build %.1f s; V=%.0f nodes, K=%.1f photons per lookup" %
                      (done, elapsed, cores, t_build, ctr["n_nodes_visited"] / max(1, ctr["n_lookups"]), ctr["n_kept"] / max(1, ctr["n_lookups"]))}, ctr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--xres", type=int, default=1280)
    ap.add_argument("--yres", type=int, default=720)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--photons", type=int, default=1000000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stats", action="store_true", help="collect gather work counters (slower)")
    ap.add_argument("--cell-scale", type=float, default=0.0, help="photon-grid cell edge multiplier (0 = library default)")
    ap.add_argument("--photon-source", choices=["shoot", "synth"], default="shoot",
                    help="shoot: device photon shooter (pvol_preprocess); synth: resampled committed map")
    ap.add_argument("--shoot-tasks", type=int, default=16384, help="virtual PhotonShootingTasks of the device shooter")
    ap.add_argument("--strong", action="store_true",
                    help="N>1: partition ONE frame's render tasks over the ranks and all-reduce the film (strong scaling) "
                         "instead of one frame per rank")
    ap.add_argument("--driver", choices=["tile", "batch"], default="tile",
                    help="tile: whole SamplerRendererTasks on the device (LD sampler, camera, Li, film; pvol_render_tasks_device); "
                         "batch: Li() only over pre-built synthetic camera rays (pvol_li_batch_device)")
    ap.add_argument("--save-image", default="", help="tile driver: write the resolved RGB film of the last step as .npy")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = importlib.import_module("cs348b-pbrt_amd")
    pvol = importlib.import_module("cs348b-pbrt_amd.pvol")
    abi, blob = pkg.abi, pkg.blob
    scene = blob.load(os.path.join(GOLD, "scene_volumescene_h.bin"))
    params = abi.params_from_blob(scene, n_volume_photons=args.photons, device=local_rank, grid_cell_scale=args.cell_scale)
    pv = pvol.PhotonVolume(params)
    pv.set_scene(abi.SceneHolder(scene))
    t_map = time.perf_counter()
    if args.photon_source == "shoot":
        # PhotonShooter::Preprocess on the device; every rank shoots the same map (same seeds): replicated, no traffic
        pv.preprocess(args.shoot_tasks)
        photons = pv.download_photons() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
        photon_note = "device shooter, %d virtual tasks" % args.shoot_tasks
    else:
        photons = synth_photons(args.photons)
        pv.upload_photons(*photons)
        photon_note = "synthetic: committed 6k-photon map of the scene resampled"
    n_photons = pv.photon_count()
    t_map = time.perf_counter() - t_map

    x0s, x1s, y0s, y1s, n_tiles = frame_tiles(args.xres, args.yres)
    if args.strong:
        mine = np.arange(rank, n_tiles, world)      # round-robin keeps every rank's tasks spread over the frame
    else:
        mine = np.arange(n_tiles)                   # weak scaling: every rank renders a whole frame of its own
    stream = torch.cuda.current_stream().cuda_stream
    if args.stats:
        pv.enable_stats(True)

    cam = film = smp = None
    if args.driver == "tile":
        cam = abi.perspective_camera(float(scene["camera.fov"][0]), args.xres, args.yres, scene["camera.c2w"])
        film = abi.make_film(args.xres, args.yres, pvol.gaussian_filter_table())
        smp = abi.make_sampler(args.xres, args.yres, args.spp, n_tiles)
        task_ids = mine.astype(np.uint32)
        n_rays = pvol.render_sample_count(smp, task_ids)
        d_pixels = torch.zeros((args.yres, args.xres, 4), dtype=torch.float32, device=dev)
        d_rgb = torch.zeros((args.yres, args.xres, 3), dtype=torch.float32, device=dev)

        def step():
            d_pixels.zero_()
            pv.render_tasks(cam, film, smp, task_ids, d_pixels.data_ptr(), None, stream)
            if args.strong and dist is not None:
                dist.all_reduce(d_pixels, op=dist.ReduceOp.SUM)   # the film reduce: 4 floats per pixel over RCCL
            pv.film_resolve(film, d_pixels.data_ptr(), d_rgb.data_ptr(), stream)
    else:
        tiles = (x0s[mine], x1s[mine], y0s[mine], y1s[mine])
        rays, counts = build_rays(torch, dev, scene, args.xres, args.yres, args.spp, tiles, seed=1234 + rank)
        n_rays = int(counts.sum())
        st = abi.make_streams(mine.astype(np.uint32), counts.astype(np.uint32))   # RNG(taskNum), samplerrenderer.cpp:73
        d_streams = torch.from_numpy(st.view(np.uint8).reshape(len(st), 32).copy()).to(dev)
        d_out = torch.zeros((n_rays, 4), dtype=torch.float32, device=dev)

        def step():
            pv.li_device(rays.data_ptr(), n_rays, d_streams.data_ptr(), len(st), abi.OUT_XYZ, d_out.data_ptr(), 0, stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    pv.kernel_time_ms(reset=True)
    if args.stats:
        pv.stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([n_rays], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_rays = int(tot.item())
    else:
        total_rays = n_rays
    kms, launches = pv.kernel_time_ms()
    kernel_name = pv.march_kernel_name()
    if args.driver == "tile":
        checksum = float(d_rgb.double().sum().item())
        if args.save_image and rank == 0:
            np.save(args.save_image, d_rgb.cpu().numpy())
    else:
        checksum = float(d_out[:, :3].double().sum().item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_rays * args.steps / dt / 1e6
        if args.driver == "tile":
            what = ("whole SamplerRendererTasks on the device: LD sampler + perspective camera + Scene::Intersect clip (pre-pass kernel), "
                    "Li (march + gather), ImageFilm::AddSample + WriteRGB; surface radiance not computed (Ls = Lvi)")
        else:
            what = "Li() only over pre-built synthetic camera rays"
        res = {
            "metric": "volumetric photon-gather throughput (camera samples through PhotonVolumeIntegrator::Li per second)",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "volumescene (homogeneous) %dx%d, %d spp, %d volume photons, nused %d, maxdist %.2f, stepsize %.2f; "
                                   "%d render tasks (MT19937 streams), %d Li() calls per step incl. filter apron" %
                                   (args.xres, args.yres, args.spp, n_photons, params.n_used, params.max_dist, params.step_size,
                                    n_tiles, total_rays),
                       "step": what,
                       "photon_map": "%s: %d photons (>= %d requested), built in %.1f s (untimed setup)" % (photon_note, n_photons, args.photons, t_map),
                       "partition": (("one frame, render tasks round-robin over %d rank(s), film all-reduced over RCCL" if args.strong
                                      else "one whole frame per rank x %d rank(s), no data-path collective") % world) + ", photon map replicated"},
            "wall_s": dt, "checksum": checksum,
        }
        # roofline of the dominant kernel (li_group_kernel here): algorithmic bytes = B_lookup x lookups, SURVEY 8(d)
        cpu, ctr = (None, None)
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is a rank-0, N=1 leg
            if args.driver == "tile":
                cpu, ctr = cpu_baseline_render(scene, params, photons, cam, film, smp)
            else:
                cpu, ctr = cpu_baseline(scene, params, photons, args.xres, args.yres)
            res["cpu_baseline"] = cpu
        stats = pv.stats() if args.stats else None
        # without the CPU leg (N > 1, --no-cpu-baseline): the figures the N=1 run of this workload measured (profiles/)
        V = (ctr["n_nodes_visited"] / max(1, ctr["n_lookups"])) if ctr else 328.6
        K = (ctr["n_kept"] / max(1, ctr["n_lookups"])) if ctr else 46.6
        steps_per_ray = (ctr["n_steps"] / max(1, ctr["n_rays"])) if ctr else 37.1
        b_lookup = 20.0 * V + 132.0 * K
        bytes_per_launch = b_lookup * steps_per_ray * n_rays + 16.0 * n_rays + 48.0 * n_rays
        achieved = bytes_per_launch / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        # HBM bytes of the same kernel from rocprofv3 PMC passes of tools/pvol_prof on this workload (profiles/),
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, scaled from bytes per Li() call
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = tj["hbm_bytes_per_ray"] * n_rays
            traffic_src = tj["source"]
        res["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                           "traffic": traffic, "traffic_source": traffic_src,
                           "hbm_measured_GBps": (traffic / (kms * 1e-3) / 1e9) if (traffic and kms > 0) else None,
                           "kernel": kernel_name, "kernel_avg_ms": kms, "kernel_launches": launches,
                           "algorithmic_bytes_per_lookup": b_lookup, "V": V, "K": K, "lookups_per_sample": steps_per_ray,
                           "note": "B_lookup = 20*V + 132*K with V, K from the reference-algorithm counters of the CPU baseline on the same inputs; "
                                   "kernel_avg_ms is the HIP-event time of the march+gather kernel alone, ms_per_step the whole step. "
                                   "frac > 1 means the kernel serves the reference algorithm's bytes from LDS/L2/Infinity Cache: 64 neighbouring "
                                   "gathers share one staged photon bucket, so the HBM traffic (`traffic`, PMC) is a small fraction of them"}
        if stats:
            res["gpu_counters"] = stats
        print(json.dumps(res))
    pv.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
