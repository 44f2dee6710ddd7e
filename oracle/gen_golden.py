#!/usr/bin/env python3
"""Regenerates tests/golden/*.bin -- TEST INFRASTRUCTURE.

Runs HERE (needs /root/reference through oracle/_ref/ref_capture).  Every *ref* fixture is
written by the reference's own objects; photon maps (inputs) come from the oracle shooter
because the reference's shooter cannot be linked in this image (see oracle/Makefile).

    make -C oracle golden
"""
import importlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import orc  # noqa: E402

pkg = importlib.import_module("cs348b-pbrt_amd")
abi, blob = pkg.abi, pkg.blob
GOLD = os.path.join(ROOT, "tests", "golden")
CAP = os.path.join(HERE, "_ref", "ref_capture")


def cap(*args):
    subprocess.check_call([CAP] + [str(a) for a in args])


def camera_rays(scene, nx, ny, rng, x0=0.0, x1=1.0, y0=0.0, y1=1.0):
    """Pinhole rays through an nx x ny lattice of the film window [x0,x1]x[y0,y1] (jittered)."""
    c2w = scene["camera.c2w"].reshape(4, 4).astype(np.float64)
    t = np.tan(np.radians(float(scene["camera.fov"][0])) / 2)
    xs = x0 + (np.arange(nx) + rng.random(nx)) / nx * (x1 - x0)
    ys = y0 + (np.arange(ny) + rng.random(ny)) / ny * (y1 - y0)
    X, Y = np.meshgrid(xs, ys)
    d = np.stack([(2 * X - 1) * t, (1 - 2 * Y) * t, np.ones_like(X)], -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d @ c2w[:3, :3].T
    o = np.tile(c2w[:3, 3], (len(d), 1))
    return o.astype(np.float32), d.astype(np.float32)


def clip_to_surfaces(oracle_ctx, o, d):
    """ray.maxt after SamplerRenderer::Li's scene->Intersect (samplerrenderer.cpp:234)."""
    import ctypes as C
    L = orc.lib()
    fp = C.POINTER(C.c_float)
    maxt = np.full(len(o), np.inf, np.float32)
    rec = np.zeros(11, np.float32)
    for i in range(len(o)):
        oi = np.ascontiguousarray(o[i]); di = np.ascontiguousarray(d[i])
        if L.orc_intersect(oracle_ctx._h, oi.ctypes.data_as(fp), di.ctypes.data_as(fp), 0.0, float("inf"), rec.ctypes.data_as(fp)):
            maxt[i] = rec[0]
    return maxt


def ray_blob(rays, streams, transmittance_only=False):
    b = {
        "rays.o": rays["o"].reshape(-1).copy(), "rays.d": rays["d"].reshape(-1).copy(),
        "rays.mint": rays["mint"].copy(), "rays.maxt": rays["maxt"].copy(), "rays.time": rays["time"].copy(),
        "rays.u": rays["scatter_u"].copy(), "rays.skip": rays["rng_skip"].copy(),
        "streams.seed": streams["seed"].copy(), "streams.first": streams["first_ray"].copy(),
        "streams.n": streams["n_rays"].copy(), "streams.start": streams["start_draw"].copy(),
    }
    if transmittance_only:
        b["transmittance_only"] = np.ones(1, np.uint32)
    return b


def make_case(name, scene_name, nx, ny, n_streams, seed, photons=None, overrides=None, window=(0, 1, 0, 1),
              extra_rays=None, transmittance_only=False):
    """Writes li_<name>.bin = inputs (rays, streams, params) + the REFERENCE's outputs."""
    rng = np.random.default_rng(seed)
    scene = blob.load(os.path.join(GOLD, "scene_%s.bin" % scene_name))
    holder = abi.SceneHolder(scene)
    over = overrides or {}
    params = abi.params_from_blob(scene, **{k: v for k, v in over.items()})
    oc = orc.Oracle(holder, params)
    o, d = camera_rays(scene, nx, ny, rng, *window)
    maxt = clip_to_surfaces(oc, o, d)
    n = len(o)
    rays = abi.make_rays(o, d, 0.0, maxt, rng.random(n).astype(np.float32))
    # the caller's draws between Li() calls (sampler + surface integrator): a few non-zero skips
    skip = np.where(rng.random(n) < 0.25, rng.integers(1, 700, n), 0).astype(np.uint32)
    rays["rng_skip"] = skip
    if extra_rays is not None:
        rays = np.concatenate([rays, extra_rays])
        n = len(rays)
    counts = np.full(n_streams, n // n_streams, np.uint32)
    counts[-1] += n - counts.sum()
    streams = abi.make_streams(np.arange(n_streams, dtype=np.uint32) + 4000 * (seed % 3), counts,
                               start_draw=rng.integers(0, 2000, n_streams).astype(np.uint64))
    tmp_rays = "/tmp/pvol_rays_%s.bin" % name
    tmp_out = "/tmp/pvol_out_%s.bin" % name
    blob.save(tmp_rays, ray_blob(rays, streams, transmittance_only))
    args = ["li", scene_name, os.path.join(GOLD, "photons_%s.bin" % photons) if photons else "-", tmp_rays, tmp_out]
    if "step_size" in over:
        args += ["stepsize", over["step_size"]]
    if "n_used" in over:
        args += ["nused", over["n_used"]]
    if "max_dist" in over:
        args += ["maxdist", over["max_dist"]]
    cap(*args)
    ref = blob.load(tmp_out)
    out = ray_blob(rays, streams, transmittance_only)
    out["params.f"] = np.array([params.step_size, params.max_dist, params.shooter_step_size], np.float32)
    out["params.nused"] = np.array([params.n_used], np.int32)
    out["ref.Lv"] = ref["Lv"]
    out["ref.T"] = ref["T"]
    out["ref.draws"] = ref["draws"]
    out["ref.next_rng"] = ref["next_rng"]
    out["ref.streams.end"] = ref["streams.end"]
    blob.save(os.path.join(GOLD, "li_%s.bin" % name), out)
    os.remove(tmp_rays)
    os.remove(tmp_out)
    lv = ref["Lv"].reshape(-1, 30)
    print("%-28s rays %4d  mean|Lv| %.4g  nonzero rays %d  draws/ray %.1f" %
          (name, n, np.abs(lv).mean(), (np.abs(lv).sum(1) > 0).sum(), ref["draws"].mean()))


def shoot(scene_name, n_photons, tag, n_tasks=1, **over):
    scene = blob.load(os.path.join(GOLD, "scene_%s.bin" % scene_name))
    holder = abi.SceneHolder(scene)
    params = abi.params_from_blob(scene, n_volume_photons=n_photons, **over)
    oc = orc.Oracle(holder, params)
    rc = oc.shoot(n_tasks, 8 if n_tasks > 1 else 1)
    assert rc == 0, rc
    P, W, A = oc.get_photons()
    st = oc.shoot_stats()
    blob.save(os.path.join(GOLD, "photons_%s.bin" % tag), {
        "p": P.reshape(-1), "wi": W.reshape(-1), "alpha": A.reshape(-1),
        "shoot_stats": np.array([st[k] for k in orc.SHOOT_STAT_NAMES], np.uint64),
        "n_tasks": np.array([n_tasks], np.uint32)})
    print("photons_%s: %d photons, %d paths" % (tag, len(P), st["paths"]))


def shoot_caustic(scene_name, n_photons, tag, n_tasks=1, keep=None, **over):
    """The caustic store of the oracle shooter's run that also made photons_<tag> (same seeds, same paths): input of the
    surface integrator's captures.  The reference's own shooter cannot be linked here (oracle/Makefile)."""
    scene = blob.load(os.path.join(GOLD, "scene_%s.bin" % scene_name))
    params = abi.params_from_blob(scene, n_volume_photons=n_photons, **over)
    oc = orc.Oracle(abi.SceneHolder(scene), params)
    oc.keep_surface_photons(True)
    assert oc.shoot(n_tasks, 1) == 0
    P, W, A, npaths = oc.surface_photons(0)
    if keep is not None:   # an input of both sides: the first `keep` photons are as good a caustic map as all of them, and smaller
        P, W, A = P[:keep], W[:keep], A[:keep]
    blob.save(os.path.join(GOLD, "caustic_%s.bin" % tag), {"p": P.reshape(-1), "wo": W.reshape(-1), "alpha": A.reshape(-1),
                                                            "n_paths": np.array([npaths], np.uint32)})
    print("caustic_%s: %d photons, %d paths" % (tag, len(P), npaths))


def render_case(tag, scene_name, photons, xres, yres, spp, ntasks, tasks=None, **over):
    """Reference SamplerRendererTask loop (LDSampler + PerspectiveCamera + Li + ImageFilm), ref_capture `render`."""
    args = ["render", scene_name, os.path.join(GOLD, "photons_%s.bin" % photons) if photons else "-",
            os.path.join(GOLD, "render_%s.bin" % tag), "xres", xres, "yres", yres, "spp", spp, "ntasks", ntasks]
    if tasks is not None:
        args += ["tasks", ",".join(str(t) for t in tasks)]
    for k, v in over.items():
        args += [k, v]
    cap(*args)


def main_surface():
    """SURVEY 8(f)-2, matte subset: the reference's PhotonIntegrator::Li (direct lighting + caustic estimate; the scene's
    indirectphotons 0 leaves the final gather without a map) composed with the volume term as SamplerRenderer::Li does."""
    shoot_caustic("volumescene_h", 6000, "vh")
    render_case("vh_surf", "volumescene_h", "vh", 32, 18, 4, 8, surface=os.path.join(GOLD, "caustic_vh.bin"))
    render_case("vh_surf64", "volumescene_h", "vh", 10, 6, 64, 4, tasks=[0, 2, 3], surface=os.path.join(GOLD, "caustic_vh.bin"))


def main_hg():
    """Row a16 (PhaseHG with g != 0, core/volume.cpp:150-154): the volumescene with `"float g" 0.6` on the Volume."""
    cap("scene", "volumescene_hg", os.path.join(GOLD, "scene_volumescene_hg.bin"))
    cap("units", "volumescene_hg", os.path.join(GOLD, "ref_units_volumescene_hg.bin"))
    shoot("volumescene_hg", 6000, "vhg")          # the shooter's scattering weight alpha *= p(wo, wi) / pdf now varies
    make_case("vhg", "volumescene_hg", 12, 10, 3, 11, photons="vhg")
    make_case("vhg_k20", "volumescene_hg", 8, 8, 2, 12, photons="vhg", overrides={"n_used": 20, "max_dist": 0.3})


def main_mesh():
    """Row f4 (more triangles than a linear scan is for): the volumescene room with a 960-triangle matte ball in the medium.
    The reference answers every hit through its BVHAccel; `units` holds 2 000 closest / any hit records (half of the rays aimed
    at the mesh), `li` and `render` the integrator and whole render tasks with shadow rays and camera rays against the mesh."""
    cap("scene", "meshroom", os.path.join(GOLD, "scene_meshroom.bin"))
    cap("units", "meshroom", os.path.join(GOLD, "ref_units_meshroom.bin"))
    shoot("meshroom", 4000, "mesh")
    make_case("mesh", "meshroom", 16, 12, 4, 21, photons="mesh")
    render_case("mesh", "meshroom", "mesh", 24, 16, 4, 6)


def main_sphere():
    """Row f3, Shape "sphere": the reference's sphere scene (a glass ball in the medium, spot + point light) -- `spherescene` is
    that file as the reference's API builds it (held against the scene-file front end), `sphereroom` adds a partial matte
    sphere under a rotation and a non-uniform scale.  Hits, BSDF samples, Li records and whole render tasks come from the
    reference's Sphere::Intersect / IntersectP through its BVHAccel."""
    cap("scene", "spherescene", os.path.join(GOLD, "scene_spherescene.bin"))
    cap("scene", "sphereroom", os.path.join(GOLD, "scene_sphereroom.bin"))
    cap("units", "sphereroom", os.path.join(GOLD, "ref_units_sphereroom.bin"))
    shoot("sphereroom", 4000, "sph")
    make_case("sph", "sphereroom", 14, 12, 3, 31, photons="sph", overrides={"n_used": 50, "step_size": 0.1})
    render_case("sph", "sphereroom", "sph", 24, 16, 4, 6, stepsize=0.1, nused=50)


def main_specular():
    """Row f2, the specular recursion (SpecularReflect / SpecularTransmit, core/integrator.cpp:177-262): camera rays that meet
    the glass prism of pinkfloyd / the glass ball of the sphere scene spawn rays through SamplerRenderer::Li, whose surface and
    volume terms draw from the same stream before the primary ray's volume term does.  Captured with the reference's own
    PhotonIntegrator; the frames are aimed at the glass (small windows of a larger frame via `tasks`)."""
    shoot_caustic("pinkfloyd", 6000, "pf", keep=3000)
    render_case("pf_surf", "pinkfloyd", "pf", 48, 48, 4, 16, tasks=[1, 2, 5, 6], surface=os.path.join(GOLD, "caustic_pf.bin"))
    shoot_caustic("sphereroom", 4000, "sph", keep=3000)
    render_case("sph_surf", "sphereroom", "sph", 32, 32, 4, 16, tasks=[5, 6, 9, 10], stepsize=0.1, nused=50,
                surface=os.path.join(GOLD, "caustic_sph.bin"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "hg":   # only the fixtures added in round 2
        return main_hg()
    if len(sys.argv) > 1 and sys.argv[1] == "mesh":
        return main_mesh()
    if len(sys.argv) > 1 and sys.argv[1] == "sphere":
        return main_sphere()
    if len(sys.argv) > 1 and sys.argv[1] == "specular":
        return main_specular()
    if len(sys.argv) > 1 and sys.argv[1] == "surface":
        return main_surface()
    os.makedirs(GOLD, exist_ok=True)
    cap("tables", os.path.join(GOLD, "ref_tables.bin"))
    for s in ["volumescene_h", "volumescene_rainbow", "volumescene_grid16", "pinkfloyd", "shootbench"]:
        cap("scene", s, os.path.join(GOLD, "scene_%s.bin" % s))
        cap("units", s, os.path.join(GOLD, "ref_units_%s.bin" % s))
    # photon maps (oracle shooter; inputs, not reference outputs)
    shoot("volumescene_h", 6000, "vh")
    shoot("pinkfloyd", 6000, "pf")
    shoot("volumescene_grid16", 4000, "grid16")
    # reference Li() records
    make_case("vh", "volumescene_h", 16, 12, 4, 1, photons="vh")
    make_case("vh_sparse", "volumescene_h", 8, 8, 2, 2, photons="vh", overrides={"max_dist": 0.12, "n_used": 50})
    make_case("vh_k500", "volumescene_h", 8, 6, 3, 3, photons="vh", overrides={"n_used": 500, "max_dist": 0.9})
    make_case("vh_nomap", "volumescene_h", 6, 6, 1, 4, photons=None)
    make_case("rainbow", "volumescene_rainbow", 12, 10, 3, 5, photons=None)
    make_case("grid16", "volumescene_grid16", 10, 8, 2, 6, photons="grid16")
    make_case("pf", "pinkfloyd", 10, 8, 4, 7, photons="pf", window=(0.25, 0.85, 0.2, 0.7))
    make_case("pf_k50", "pinkfloyd", 8, 8, 2, 8, photons="pf", overrides={"n_used": 50, "max_dist": 0.25},
              window=(0.3, 0.8, 0.25, 0.65))
    # Transmittance() records (sample == NULL path)
    make_case("trans_vh", "volumescene_h", 8, 8, 2, 9, photons=None, transmittance_only=True)
    make_case("trans_grid16", "volumescene_grid16", 8, 8, 2, 10, photons=None, transmittance_only=True)
    # whole render tasks: sampler + camera + Li + film
    render_case("vh", "volumescene_h", "vh", 32, 18, 4, 8)
    render_case("vh64", "volumescene_h", "vh", 10, 6, 64, 4, tasks=[0, 2, 3])
    render_case("grid16", "volumescene_grid16", "grid16", 16, 10, 2, 4)
    render_case("pf", "pinkfloyd", "pf", 16, 16, 4, 4, nused=50, maxdist=0.25)
    main_hg()
    main_surface()


if __name__ == "__main__":
    main()
