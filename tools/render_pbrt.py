#!/usr/bin/env python3
"""Renders a .pbrt scene file through the device pipeline: scene-file front end (cs348b-pbrt_amd/pbrt_scene.py) -> photon shooter
-> SamplerRendererTasks on the device (LD sampler, camera, surface PhotonIntegrator where it applies, PhotonVolumeIntegrator,
image film) -> RGB.

    python tools/render_pbrt.py SCENE.pbrt OUT.pfm [--xres N --yres N --spp N --photons N --shoot-tasks N --no-surface]

The file's own Film / Sampler / integrator parameters are used unless overridden.  The surface integrator (direct lighting +
caustic estimate on matte surfaces) is switched on when the scene asks for "photonmap" and the device path
covers it (matte and glass surfaces -- the specular recursion included --, homogeneous isotropic medium, no indirect map); otherwise
Ls = 0 and the image holds the volume term alone -- the tool says which."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def render_scene_file(path, xres=None, yres=None, spp=None, photons=None, shoot_tasks=2048, surface=True, log=print, caustic_photons=None):
    import torch
    pkg = importlib.import_module("cs348b-pbrt_amd")
    pvol = importlib.import_module("cs348b-pbrt_amd.pvol")
    ps = importlib.import_module("cs348b-pbrt_amd.pbrt_scene")
    abi = pkg.abi
    scene = ps.load(path)
    xres, yres, spp = xres or int(scene["film"][0]), yres or int(scene["film"][1]), spp or int(scene["film"][2])
    spp = 1 << max(0, int(spp - 1).bit_length())   # LDSampler rounds up to a power of two (samplers/lowdiscrepancy.cpp:41-47)
    over = {"keep_surface_photons": 1}
    if photons:
        over["n_volume_photons"] = int(photons)
    if caustic_photons is not None:
        over["n_caustic_photons"] = int(caustic_photons)
    params = abi.params_from_blob(scene, **over)
    pv = pvol.PhotonVolume(params)
    try:
        holder = abi.SceneHolder(scene)
        pv.set_scene(holder)
        if int(scene["lights.kind"].size):
            pv.preprocess(shoot_tasks)
        st = pv.shoot_stats()
        log("photon map: %d volume, %d caustic photons from %d paths" % (st["stored_volume"], st["stored_caustic"], st["paths"]))
        used_surface = False
        if surface and scene["surf.name"] == "photonmap":
            try:
                pv.set_surface_integrator(int(scene["surf.params.i"][0]), float(scene["surf.params.f"][0]), int(scene["surf.params.i"][1]),
                                          bool(scene["surf.params.i"][2]), from_preprocess=True)
                used_surface = True
            except pvol.PvolError as e:
                log("surface integrator not applied (%s): the image holds the volume term only" % e)
        import bench   # the reference's task count: max(32 x cores, pixels / 256) rounded up to a power of two (samplerrenderer.cpp:206-208)
        n_tiles = int(bench.frame_tiles(xres, yres)[4])
        cam = abi.perspective_camera(float(scene["camera.fov"][0]), xres, yres, scene["camera.c2w"])
        film = abi.make_film(xres, yres, pvol.gaussian_filter_table())
        smp = abi.make_sampler(xres, yres, spp, n_tiles)
        ids = np.arange(n_tiles, dtype=np.uint32)
        dev = torch.device("cuda:0")
        px = torch.zeros((yres, xres, 4), dtype=torch.float32, device=dev)
        rgb = torch.zeros((yres, xres, 3), dtype=torch.float32, device=dev)
        try:
            pv.render_tasks(cam, film, smp, ids, px.data_ptr())
        except pvol.PvolError as e:
            if not used_surface:
                raise
            log("surface integrator not applied (%s): the image holds the volume term only" % e)
            pv.set_surface_integrator(off=True)
            used_surface = False
            px.zero_()
            pv.render_tasks(cam, film, smp, ids, px.data_ptr())
        pv.film_resolve(film, px.data_ptr(), rgb.data_ptr())
        torch.cuda.synchronize()
        pv.check_errors()
        return rgb.cpu().numpy(), {"xres": xres, "yres": yres, "spp": spp, "surface_integrator": used_surface, "kernel": pv.march_kernel_name(),
                                   "photons": int(st["stored_volume"]), "caustic_photons": int(st["stored_caustic"])}
    finally:
        pv.close()


def write_pfm(path, rgb):
    h, w, _ = rgb.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(rgb[::-1], dtype="<f4").tobytes())   # PFM stores the bottom row first


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("out")
    ap.add_argument("--xres", type=int)
    ap.add_argument("--yres", type=int)
    ap.add_argument("--spp", type=int)
    ap.add_argument("--photons", type=int)
    ap.add_argument("--shoot-tasks", type=int, default=2048)
    ap.add_argument("--no-surface", action="store_true")
    a = ap.parse_args()
    img, info = render_scene_file(a.scene, a.xres, a.yres, a.spp, a.photons, a.shoot_tasks, not a.no_surface)
    write_pfm(a.out, img)
    print(info, "mean rgb", img.mean(axis=(0, 1)))
