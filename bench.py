#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: Msamples/s of the volumetric photon-mapping hot path
(ray march + k-NN photon gather, PhotonVolumeIntegrator::Li) on config[1]:
projectScene/volumescene (homogeneous variant) at 1280x720, 256 spp, 1 M volume photons.

One "step" = one pass of the hot path over the whole frame's camera samples (incl. the filter apron:
1285 x 725 x 256 = 238.5 M Li() calls), everything resident in HBM: the default `--driver tile` runs whole
SamplerRendererTasks on the device (LD sampler + camera pre-pass, Li, film), `--driver batch` times Li() alone
over pre-built rays.  N > 1 (default): ONE frame, its render tasks (one MT19937 stream each) partitioned round-robin
over the ranks, photon map replicated, one RCCL all-reduce of the film per step inside the timed region (strong
scaling, the north_star's split); `--weak` renders one whole frame per rank instead (replicas, no collective).
value = samples of all ranks / max-over-ranks time.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N --dry-run                   (prints the partition; needs no GPU)
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


def cpu_baseline_render(scene, params, photons, cam, film, smp, budget_s=20.0, surface=None):
    """CPU baseline of the SAME pipeline (tile driver): the oracle's SamplerRendererTask loop (LD sampler, camera,
    Li with the kd-tree gather, film) on random whole render tasks of the same frame, all host cores, ~20 s.
    The first batch keeps its per-sample records: the GPU renders the same tasks afterwards and the two are compared
    (`parity` in the output line)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    pkg = importlib.import_module("cs348b-pbrt_amd")
    abi = pkg.abi
    cores = max(1, min(os.cpu_count() or 1, 16))   # the GPU box grants a 16-core CPU share per GPU
    o = orc.Oracle(abi.SceneHolder(scene), params)
    t0 = time.time()
    o.set_photons(*photons)
    t_build = time.time() - t0
    if surface is not None:   # --surface: the same PhotonIntegrator in front, fed with the device shooter's caustic store
        o.set_surface_integrator(*surface)
    o.counters(reset=True)
    rng = np.random.default_rng(7)
    n_tasks = smp.n_tasks
    # bounded sample: the oracle runs one render task per thread (a task is sequential: one RNG stream), so a batch needs at
    # least `cores` tasks to use the cores it reports; small tasks are batched further up to ~0.25 M samples
    per_task = (film.x_resolution + 5) * (film.y_resolution + 5) * smp.pixel_samples / n_tasks
    batch = int(max(1, min(n_tasks, max(cores, round(250000.0 / per_task)))))
    done, elapsed = 0, 0.0
    first = None
    while elapsed < budget_s:
        pick = rng.choice(n_tasks, min(batch, n_tasks), replace=False).astype(np.uint32)
        t0 = time.time()
        r = orc.render_tasks(o, cam, film, smp, pick, records=first is None, n_threads=cores)
        dt = time.time() - t0
        if first is None:
            first = {"tasks": pick, "xyzT": r["xyzT"], "end_draws": r["end_draws"], "rays": r["rays"]}
        done += r["n_samples"]
        elapsed += dt
        if dt < 3.0:
            batch *= 2
    ctr = o.counters()
    return {"value": done / elapsed / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d camera samples = random whole render tasks of the same frame through the oracle's SamplerRendererTask loop "
                      "(LDSampler, camera, Li with the kd-tree gather of core/kdtree.h, ImageFilm; same scene/photon map/params), "
                      "%.1f s on %d threads; kd build %.1f s; V=%.0f nodes, K=%.1f photons per lookup" %
                      (done, elapsed, cores, t_build, ctr["n_nodes_visited"] / max(1, ctr["n_lookups"]), ctr["n_kept"] / max(1, ctr["n_lookups"]))}, ctr, first


def gpu_parity(torch, dev, pv, abi, cam, film, smp, first, stream):
    """The GPU renders the tasks whose oracle records the CPU leg kept: per-sample XYZ (rel. L2 per sample and per pixel =
    mean over the pixel's samples), RNG stream end positions (exact)."""
    tasks = np.asarray(first["tasks"], np.uint32)
    n = len(first["xyzT"])
    px = torch.zeros((film.y_resolution, film.x_resolution, 4), dtype=torch.float32, device=dev)
    xyz = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    streams = torch.zeros((len(tasks), 32), dtype=torch.uint8, device=dev)
    dbg = abi.RenderDebug(0, 0, xyz.data_ptr(), streams.data_ptr())
    pv.render_tasks(cam, film, smp, tasks, px.data_ptr(), dbg, stream)
    torch.cuda.synchronize()
    got = xyz.cpu().numpy().astype(np.float64)
    ref = first["xyzT"].astype(np.float64)
    end = streams.cpu().numpy().view(abi.STREAM_DTYPE).reshape(-1)["end_draw"]
    scale = max(float(np.abs(ref[:, :3]).max()), 1e-30)
    err = np.linalg.norm(got[:, :3] - ref[:, :3], axis=1) / np.maximum(np.linalg.norm(ref[:, :3], axis=1), 1e-6 * scale)
    spp = smp.pixel_samples
    gp, rp = got[:, :3].reshape(-1, spp, 3).mean(1), ref[:, :3].reshape(-1, spp, 3).mean(1)
    perr = np.linalg.norm(gp - rp, axis=1) / np.maximum(np.linalg.norm(rp, axis=1), 1e-6 * scale)
    # bar (north_star): <= 1e-4 relative L2 per PIXEL, stream positions exact.  Per sample the figure is reported too: it
    # exceeds 1e-4 only where two photons tie exactly for the k-th place of a lookup (the reference keeps the one its kd-tree
    # traversal met first; see tests/test_gpu_group.py::test_headline_shape_256spp_on_a_million_photon_map)
    return {"samples": int(n), "tasks": int(len(tasks)), "max_rel_l2_per_pixel": float(perr.max()), "max_rel_l2_per_sample": float(err.max()),
            "samples_above_1e-4": int((err > 1e-4).sum()),
            "rng_end_positions_equal": bool((end == first["end_draws"]).all()), "tolerance_per_pixel": 1e-4,
            "ok": bool(perr.max() <= 1e-4 and (end == first["end_draws"]).all())}


def spawn_ranks(n):
    """`bench.py --gpus N` started plainly: launch N ranks (one per GPU) as a CHILD torchrun job before this process touches
    the GPU, and pass its output and exit code through.  Never an exec: the box forbids replacing a process that may own a GPU."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def partition_tasks(n_tiles, rank, world, weak):
    """north_star: the frame's render tasks (pixel tiles) are partitioned over the ranks, round-robin so every rank's share is
    spread over the frame; --weak gives every rank a whole frame of its own instead."""
    if weak:
        return np.arange(n_tiles)
    pvol = importlib.import_module("cs348b-pbrt_amd.pvol")   # the library's own partition (pvol_partition_tasks): what the C++ binding uses
    return pvol.partition_tasks(int(n_tiles), int(rank), int(world)).astype(np.int64)


def emulate_ranks(args, torch, pv, pvol, abi, cam, film, smp, n_tiles, d_pixels, d_rgb, stream, n_photons):
    """What ONE rank of an N-GPU strong-scaling run does, measured on the one GPU at hand: rank r's round-robin task list
    rendered alone, per-phase device time from HIP events (pvol_get_phase_ms).  Projected N-GPU step = slowest emulated rank
    + the film all-reduce (16 B per pixel; ring over xGMI at ~100 GB/s effective is < 1 ms at 720p and is ADDED as a constant
    1.0 ms).  Efficiency = T(1) / (N * T(N)).  Not a measurement of N GPUs: no second card, no RCCL call."""
    assert args.driver == "tile" and cam is not None
    pv.enable_phase_timing(True)
    sizes = [1] + [int(x) for x in args.emulate_ranks.split(",") if int(x) > 1]
    out = {"emulation": True, "note": "per-rank task lists rendered one at a time on ONE GPU; projection, not a scaling measurement",
           "workload": "volumescene (homogeneous) %dx%d, %d spp, %d photons, %d render tasks" % (args.xres, args.yres, args.spp, n_photons, n_tiles),
           "reduce_ms_assumed": 1.0, "worlds": []}
    t1 = None
    for N in sizes:
        ranks = sorted(set([0, N // 2, N - 1]))
        rows = []
        for r in ranks:
            ids = partition_tasks(n_tiles, r, N, False).astype(np.uint32)
            n_rays = pvol.render_sample_count(smp, ids)

            def step():
                d_pixels.zero_()
                pv.render_tasks(cam, film, smp, ids, d_pixels.data_ptr(), None, stream)
                pv.film_resolve(film, d_pixels.data_ptr(), d_rgb.data_ptr(), stream)
            for _ in range(max(1, args.warmup)):
                step()
            torch.cuda.synchronize()
            pv.phase_ms(reset=True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps * 1e3
            ph = {k: v / args.steps for k, v in pv.phase_ms(reset=True).items()}
            pv.check_errors()
            rows.append({"rank": r, "tasks": int(len(ids)), "samples": int(n_rays), "step_ms": dt, "phases_ms": ph})
        slow = max(x["step_ms"] for x in rows)
        proj = slow + (out["reduce_ms_assumed"] if N > 1 else 0.0)
        if N == 1:
            t1 = proj
        total = pvol.render_sample_count(smp, np.arange(n_tiles, dtype=np.uint32))
        out["worlds"].append({"n_gpus": N, "ranks": rows, "projected_step_ms": proj, "projected_Msamples_per_s": total / proj / 1e3,
                              "projected_efficiency": t1 / (N * proj)})
    print(json.dumps(out))
    pv.close()
    return 0


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
    ap.add_argument("--stats", action="store_true", help="collect gather work counters inside the timed steps (slower)")
    ap.add_argument("--cell-scale", type=float, default=0.0, help="photon-grid cell edge multiplier (0 = library default)")
    ap.add_argument("--photon-source", choices=["shoot", "synth"], default="shoot",
                    help="shoot: device photon shooter (pvol_preprocess); synth: resampled committed map")
    ap.add_argument("--shoot-tasks", type=int, default=16384, help="virtual PhotonShootingTasks of the device shooter")
    ap.add_argument("--weak", action="store_true",
                    help="N>1: one whole frame per rank, no data-path collective (replicas) instead of the default: ONE frame's "
                         "render tasks partitioned over the ranks + one RCCL all-reduce of the film (the north_star's split)")
    ap.add_argument("--strong", action="store_true", help="(default for N>1; kept for old command lines)")
    ap.add_argument("--dry-run", action="store_true", help="print the rank/task partition for --gpus N and exit (no GPU needed)")
    ap.add_argument("--driver", choices=["tile", "batch"], default="tile",
                    help="tile: whole SamplerRendererTasks on the device (LD sampler, camera, Li, film; pvol_render_tasks_device); "
                         "batch: Li() only over pre-built synthetic camera rays (pvol_li_batch_device)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks on ONE card (all on cuda:0, gloo for the collectives, the film staged through the host): "
                         "exercises the partition, the film reduce and the rank reductions where no multi-GPU node is at hand; not a measurement")
    ap.add_argument("--emulate-ranks", default="",
                    help="comma list of world sizes N (e.g. 2,4,8): on ONE GPU render the task list rank r of N would get (r = 0, N/2, N-1), "
                         "time the step and its phases (tile pre-pass / march + gather / film), and print the projected N-GPU step time and "
                         "strong-scaling efficiency.  An emulation on one card, not a scaling measurement")
    ap.add_argument("--surface", action="store_true",
                    help="tile driver: also run the scene's SurfaceIntegrator \"photonmap\" (nused 300, maxdist .15, 5000 caustic photons kept by the "
                         "device shooter, indirectphotons 0) in front of the volume term: Ls = T * Lsurface + Lvi as SamplerRenderer::Li composes it")
    ap.add_argument("--save-image", default="", help="tile driver: write the resolved RGB film of the last step as .npy")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if args.dry_run:
        n_tiles = frame_tiles(args.xres, args.yres)[4]
        parts = [partition_tasks(n_tiles, r, args.gpus, args.weak) for r in range(args.gpus)]
        print(json.dumps({"n_gpus": args.gpus, "scaling": "weak" if args.weak else ("strong" if args.gpus > 1 else "weak"),
                          "render_tasks": int(n_tiles), "tasks_per_rank": [int(len(p)) for p in parts],
                          "first_tasks_per_rank": [[int(t) for t in p[:4]] for p in parts],
                          "collective": None if (args.weak or args.gpus == 1) else "all_reduce(sum) of the film, %d bytes per rank" % (args.xres * args.yres * 16)}))
        return 0
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args.gpus)       # before any GPU call in this process
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch as `python bench.py --gpus N` or\n  python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...\n" % (args.gpus, world))
        return 2

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    rehearse = bool(args.rehearse_one_gpu) and world > 1
    if rehearse:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    strong = world > 1 and not args.weak

    pkg = importlib.import_module("cs348b-pbrt_amd")
    pvol = importlib.import_module("cs348b-pbrt_amd.pvol")
    abi, blob = pkg.abi, pkg.blob
    scene = blob.load(os.path.join(GOLD, "scene_volumescene_h.bin"))
    params = abi.params_from_blob(scene, n_volume_photons=args.photons, device=local_rank, grid_cell_scale=args.cell_scale,
                                  keep_surface_photons=1 if args.surface else 0)
    pv = pvol.PhotonVolume(params)
    pv.set_scene(abi.SceneHolder(scene))
    t_map = time.perf_counter()
    shoot_s = build_s = None
    if args.photon_source == "shoot":
        # PhotonShooter::Preprocess on the device; every rank shoots the same map (same seeds): replicated, no traffic
        pv.preprocess(args.shoot_tasks)
        shoot_s, build_s = pv.preprocess_times()
        photons = pv.download_photons() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
        photon_note = "device shooter, %d virtual tasks" % args.shoot_tasks
    else:
        photons = synth_photons(args.photons)
        pv.upload_photons(*photons)
        photon_note = "synthetic: committed 6k-photon map of the scene resampled"
    n_photons = pv.photon_count()
    if args.surface:
        if args.photon_source != "shoot" or args.driver != "tile":
            raise SystemExit("--surface needs the device shooter's caustic store and the tile driver")
        pv.set_surface_integrator(300, 0.15, 5, True, from_preprocess=True)   # projectScene/volumescene_png.pbrt:8-12
    torch.cuda.synchronize()
    t_map = time.perf_counter() - t_map

    x0s, x1s, y0s, y1s, n_tiles = frame_tiles(args.xres, args.yres)
    mine = partition_tasks(n_tiles, rank, world, not strong)
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
            if strong and rehearse:
                host = d_pixels.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                d_pixels.copy_(host)
            elif strong:
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

    if args.emulate_ranks:
        return emulate_ranks(args, torch, pv, pvol, abi, cam, film, smp, n_tiles, d_pixels, d_rgb, stream, n_photons)

    for _ in range(args.warmup):
        step()
    barrier()
    pv.kernel_time_ms(reset=True)
    pv.stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    pv.check_errors()
    if dist is not None:
        rdev = torch.device("cpu") if rehearse else dev
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([n_rays], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_rays = int(tot.item())
    else:
        total_rays = n_rays
    kms, launches = pv.kernel_time_ms()
    kernel_name = pv.march_kernel_name()
    work = pv.stats()   # n_rays / n_steps are counted by the kernels in every build (one atomic per wave)
    if args.driver == "tile":
        checksum = float(d_rgb.double().sum().item())
        if args.save_image and rank == 0:
            np.save(args.save_image, d_rgb.cpu().numpy())
    else:
        checksum = float(d_out[:, :3].double().sum().item())

    rc = 0
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_rays * args.steps / dt / 1e6
        if args.driver == "tile":
            what = ("whole SamplerRendererTasks on the device: LD sampler + perspective camera + Scene::Intersect clip (pre-pass kernel), "
                    "Li (march + gather), ImageFilm::AddSample + WriteRGB; " +
                    ("surface integrator ON: PhotonIntegrator::Li on the matte walls (direct lighting + caustic estimate), Ls = T * Lsurface + Lvi"
                     if args.surface else "surface radiance not computed (Ls = Lvi)"))
        else:
            what = "Li() only over pre-built synthetic camera rays"
        if world == 1:
            part = "one frame on one GPU"
        elif strong:
            part = "ONE frame: %d render tasks round-robin over %d ranks (%d each), one RCCL all-reduce of the film (%d B) per step inside the timed region" % (
                n_tiles, world, len(mine), args.xres * args.yres * 16)
        else:
            part = "one whole frame per rank x %d ranks (replicas), no data-path collective" % world
        res = {
            "metric": "volumetric photon-gather throughput (camera samples through PhotonVolumeIntegrator::Li per second)",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "volumescene (homogeneous) %dx%d, %d spp, %d volume photons, nused %d, maxdist %.2f, stepsize %.2f; "
                                   "%d render tasks (MT19937 streams), %d Li() calls per step incl. filter apron" %
                                   (args.xres, args.yres, args.spp, n_photons, params.n_used, params.max_dist, params.step_size,
                                    n_tiles, total_rays),
                       "step": what,
                       "photon_map": "%s: %d photons (>= %d requested), built in %.1f s (outside the timed steps; see end_to_end)" % (photon_note, n_photons, args.photons, t_map),
                       "partition": part + ", photon map replicated" +
                                    (" -- REHEARSAL: all ranks on one card, gloo collectives, film staged through the host; not a measurement" if rehearse else "")},
            "wall_s": dt, "checksum": checksum,
            # BASELINE.json's metric is Msamples/s AND wall-clock: one frame end to end = photon shoot + search-structure build +
            # one render step (sampler/camera pre-pass, march + gather, film splat, film reduce, resolve)
            "end_to_end": {"shoot_s": shoot_s, "grid_build_s": build_s, "photon_map_total_s": t_map, "render_step_s": ms_per_step * 1e-3,
                           "total_s": t_map + ms_per_step * 1e-3,
                           "note": "photon map is shot once per frame and replicated per rank; render_step_s includes the film all-reduce when N > 1"},
        }
        cpu, ctr, first = (None, None, None)
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is a rank-0, N=1 leg
            if args.driver == "tile":
                surf = None
                if args.surface:
                    cp, cwo, ca, cpaths = pv.surface_photons(0)
                    surf = (300, 0.15, True, (cp, cwo, ca), int(cpaths))
                cpu, ctr, first = cpu_baseline_render(scene, params, photons, cam, film, smp, surface=surf)
            else:
                cpu, ctr = cpu_baseline(scene, params, photons, args.xres, args.yres)
            res["cpu_baseline"] = cpu
        if first is not None:
            res["parity"] = gpu_parity(torch, dev, pv, abi, cam, film, smp, first, stream)
            res["parity_max_rel_l2"] = res["parity"]["max_rel_l2_per_pixel"]
            if not res["parity"]["ok"]:
                rc = 3
        # ---- roofline of the dominant kernel.  What binds it is the vector ALU, not HBM (measured traffic is ~10 % of the 8 TB/s
        # roof: the reference algorithm's bytes are served from one LDS bucket per 64 lookups), so the fraction is the VALU pipe's
        # busy time: 4 x SQ_ACTIVE_INST_VALU (the counter ticks once per four cycles a SIMD's vector pipe executes) against
        # 1024 SIMDs x 2.4 GHz.  (Rounds 1-2 quoted the ISSUE rate, instructions / (SIMD-cycles / 2): it falls when instructions are
        # removed, so it is kept as a secondary field only.)  Counters per Li() call come from the rocprofv3 PMC passes of
        # tools/pvol_prof on this kernel (profiles/).
        steps_per_ray = work["n_steps"] / max(1, work["n_rays"]) if work["n_rays"] else None
        pmc = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            pmc = json.load(open(tpath))
        peak_busy = 1024 * 2.4e9 / 1e9          # G SIMD-cycles / s
        peak_issue = 1024 * 2.4e9 / 2 / 1e9   # G wave-instructions / s
        roof = {"bound": "valu", "peak": peak_busy, "unit": "G SIMD-cycles/s", "kernel": kernel_name, "kernel_avg_ms": kms, "kernel_launches": launches,
                "lookups_per_sample": steps_per_ray, "lookups_source": "device counter (n_steps / n_rays of the timed launches)"}
        if pmc and kms > 0 and pmc.get("kernel") == kernel_name and pmc.get("valu_active_quadcycles_per_ray"):
            roof["achieved"] = 4.0 * pmc["valu_active_quadcycles_per_ray"] * n_rays / (kms * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / peak_busy
            roof["valu_issue_Gwaveinst_per_s"] = pmc["valu_insts_per_ray"] * n_rays / (kms * 1e-3) / 1e9
            roof["valu_issue_frac"] = roof["valu_issue_Gwaveinst_per_s"] / peak_issue
            if pmc.get("mfma_busy_cycles_per_ray"):
                roof["mfma_pipe_busy_frac"] = pmc["mfma_busy_cycles_per_ray"] * n_rays / (kms * 1e-3) / 1e9 / peak_busy
            roof["traffic"] = pmc["hbm_bytes_per_ray"] * n_rays
            roof["hbm_measured_GBps"] = roof["traffic"] / (kms * 1e-3) / 1e9
            roof["hbm_frac_of_8TBps"] = roof["hbm_measured_GBps"] / 8000.0
            roof["pmc_source"] = pmc["source"]
        else:
            roof.update({"achieved": None, "frac": None, "traffic": None,
                         "pmc_source": "no PMC summary for %s under profiles/ (run tools/run_profiles.sh)" % kernel_name})
        if ctr and steps_per_ray and kms > 0:
            # SURVEY 8(d)'s accounting, kept as a secondary figure: bytes the REFERENCE's kd-tree traversal would touch
            V = ctr["n_nodes_visited"] / max(1, ctr["n_lookups"])
            K = ctr["n_kept"] / max(1, ctr["n_lookups"])
            b_lookup = 20.0 * V + 132.0 * K
            alg = (b_lookup * steps_per_ray + 64.0) * n_rays
            roof["achieved_algorithmic_GBps"] = alg / (kms * 1e-3) / 1e9
            roof["algorithmic_bytes_per_lookup"] = b_lookup
            roof["V"], roof["K"] = V, K
            roof["algorithmic_note"] = ("20*V + 132*K bytes per lookup with V, K from the oracle's kd-tree counters on the same inputs; not a "
                                        "fraction of any roof: this design does not move those bytes")
        res["roofline"] = roof
        if args.stats:
            res["gpu_counters"] = work
        print(json.dumps(res))
    pv.close()
    if dist is not None:
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
