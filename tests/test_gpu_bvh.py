"""Row f4 of SURVEY 8: scenes with more triangles than the linear scan is for.  pvol_set_scene builds a linear BVH on the
device (csrc/pvol_bvh.hip) and every closest / any hit of the path walks it (csrc/pvol_bvh_dev.h).  What a traversal returns
must not depend on the tree: the bar is BIT-EXACT agreement with the oracle's linear scan over all triangles (which the
reference's own BVHAccel records pin on the meshroom scene, tests/test_oracle_vs_reference.py) -- clipped camera rays through
the tile driver for the closest hit, Li() radiance and RNG draw counts for the shadow rays.
The golden meshroom cases (Li records, whole render tasks, the shooter) run in test_gpu_parity / test_gpu_render /
test_gpu_shooter / test_gpu_group through their case tables."""
import importlib

import numpy as np
import pytest

from conftest import abi, load_photons, load_scene, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pvol():
    m = importlib.import_module("cs348b-pbrt_amd.pvol")
    assert m.lib().pvol_device_count() >= 1
    return m


def bumpy_ball(nu, nv, centre, radius, seed):
    """2 nu nv - 2 nu world-space triangles of a ball with a rippled surface (self-shadowing, grazing hits, thin slivers at the poles)."""
    rng = np.random.default_rng(seed)
    th = np.pi * np.arange(nv + 1) / nv
    ph = 2 * np.pi * np.arange(nu) / nu
    T, P = np.meshgrid(th, ph, indexing="ij")
    r = radius * (1 + 0.12 * np.sin(9 * T) * np.sin(7 * P) + 0.01 * rng.random(T.shape))
    V = np.stack([centre[0] + r * np.sin(T) * np.cos(P), centre[1] + r * np.cos(T), centre[2] + r * np.sin(T) * np.sin(P)], -1)
    V = V.astype(np.float32).reshape(-1, 3)
    j, i = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a, b = j * nu + i, j * nu + (i + 1) % nu
    c, d = (j + 1) * nu + (i + 1) % nu, (j + 1) * nu + i
    t1 = np.stack([a, b, c], -1)[j > 0]
    t2 = np.stack([a, c, d], -1)[j < nv - 1]
    idx = np.concatenate([t1, t2])
    return V[idx].reshape(-1, 9)


def big_scene(nu, nv):
    s = dict(load_scene("meshroom"))
    walls = s["tris.p"].reshape(-1, 9)[:6]
    ball = bumpy_ball(nu, nv, (0.5, 1.3, 4.0), 0.9, 5)
    s["tris.p"] = np.concatenate([walls, ball]).astype(np.float32).reshape(-1)
    n = 6 + len(ball)
    s["tris.material"] = np.concatenate([np.zeros(6, np.int32), np.ones(len(ball), np.int32)])
    s["tris.flip"] = np.zeros(n, np.int32)
    return s, n


def test_hierarchy_is_built_only_for_large_scenes(pvol):
    for name, want in [("volumescene_h", 0), ("meshroom", 966)]:
        s = load_scene(name)
        pv = pvol.PhotonVolume(abi.params_from_blob(s))
        pv.set_scene(abi.SceneHolder(s))
        n, ms = pv.accel_info()
        assert n == want
        assert (ms > 0) == (want > 0)
        pv.close()


def test_closest_and_any_hit_equal_the_linear_scan_on_37k_triangles(pvol, orc):
    from test_gpu_group import _camera_batch
    from test_gpu_render import _render
    import torch
    s, n_tris = big_scene(192, 96)
    assert n_tris > 36000
    p = abi.params_from_blob(s)
    h = abi.SceneHolder(s)
    pv = pvol.PhotonVolume(p)
    pv.set_scene(h)
    n, ms = pv.accel_info()
    assert n == n_tris
    print("LBVH over %d triangles built in %.3f ms" % (n, ms))
    pv.upload_photons(*load_photons("mesh"))
    # closest hit: the camera rays of whole render tasks, clipped by the tile driver on the device / by the oracle's scan
    # (the oracle's tile driver also marches every ray -- 34 shadow rays x 36 k triangles each -- hence the small frame)
    xres, yres, spp, n_tasks = 40, 24, 2, 4
    tasks = np.arange(n_tasks, dtype=np.uint32)
    cam = abi.perspective_camera(float(s["camera.fov"][0]), xres, yres, s["camera.c2w"])
    film = abi.make_film(xres, yres, orc.gaussian_filter_table())
    smp = abi.make_sampler(xres, yres, spp, n_tasks)
    rays = orc.render_tasks(orc.Oracle(h, p), cam, film, smp, tasks, n_threads=8)["rays"]
    r = _render(torch, pv, cam, film, smp, tasks, len(rays))
    assert (r["rays"]["d"] == rays["d"]).all()
    bare, _ = _camera_batch(orc, load_scene("volumescene_h"), xres, yres, spp, n_tasks, tasks)   # the room without the ball
    assert (bare["maxt"] != rays["maxt"]).sum() > 200    # the ball fills a good part of the frame
    assert (r["rays"]["maxt"] == rays["maxt"]).all()    # bit for bit, misses (inf) included
    # any hit: Li() of a subset -- every march step's shadow ray walks the hierarchy; draw counts depend on each outcome
    sub = np.sort(np.random.default_rng(3).choice(len(rays), 384, replace=False))
    rs = rays[sub].copy()
    rs["rng_skip"] = 0
    st = abi.make_streams(np.array([77], np.uint32), np.array([len(rs)], np.uint32))
    got, gd = pv.li(rs, st.copy())
    o = orc.Oracle(h, p)
    o.set_photons(*load_photons("mesh"))
    ref, rd = o.li_batch(rs, st.copy(), n_threads=8)
    assert (gd == rd).all()
    assert rel_l2(got[:, :30], ref[:, :30], floor=1e-6 * float(np.abs(ref[:, :30]).max())).max() <= 1e-4
    pv.close()


def _edge_scene(kind):
    """Triangle sets that stress the build: `dup` = 120 copies of one triangle (identical Morton codes: the radix tree is over the
    index bits alone, every hit a tie in t) among the walls; `flat` = degenerate (zero-area, collinear, repeated-vertex)
    triangles between real ones; `just_over` = 65 triangles, the smallest scene that takes the hierarchy."""
    s = dict(load_scene("meshroom"))
    walls = s["tris.p"].reshape(-1, 9)[:6]
    rng = np.random.default_rng(17)
    if kind == "dup":
        one = np.array([[0.0, 0.2, 4.0, 1.0, 0.2, 4.0, 0.5, 1.4, 4.2]], np.float32)
        extra = np.repeat(one, 120, axis=0)
    elif kind == "flat":
        good = bumpy_ball(8, 6, (0.5, 1.3, 4.0), 0.8, 3)
        p = rng.random((40, 3)).astype(np.float32) * 2
        bad = np.concatenate([np.concatenate([p, p, p], 1),                                   # three equal vertices
                              np.concatenate([p, p + 1, p + 2], 1),                            # collinear
                              np.concatenate([p, p, p + 0.5], 1)]).astype(np.float32)          # a repeated vertex
        extra = np.concatenate([good, bad])
        extra = extra[rng.permutation(len(extra))]
    elif kind == "with_spheres":   # hierarchy AND analytic spheres in one scene (closest hit takes the nearer of the two kinds)
        extra = bumpy_ball(24, 12, (1.6, 1.0, 4.6), 0.5, 3)
        sp = load_scene("sphereroom")
        nm = len(s["mats.kind"])
        for k in ("kind", "kd", "kr", "kt", "ior", "vn"):
            s["mats." + k] = np.concatenate([s["mats." + k], sp["mats." + k]])
        for k in ("o2w", "w2o", "f", "flip"):
            s["spheres." + k] = sp["spheres." + k]
        s["spheres.material"] = sp["spheres.material"] + nm
    else:
        extra = bumpy_ball(8, 5, (0.5, 1.3, 4.0), 0.8, 3)[:59]
    tris = np.concatenate([walls, extra]).astype(np.float32)
    s["tris.p"] = tris.reshape(-1)
    s["tris.material"] = np.concatenate([np.zeros(6, np.int32), np.ones(len(extra), np.int32)])
    s["tris.flip"] = np.zeros(len(tris), np.int32)
    return s, len(tris)


@pytest.mark.parametrize("kind", ["dup", "flat", "just_over", "with_spheres"])
def test_hierarchy_edge_cases_equal_the_linear_scan(pvol, orc, kind):
    from test_gpu_render import _render
    import torch
    s, n_tris = _edge_scene(kind)
    assert n_tris > 64
    p = abi.params_from_blob(s)
    h = abi.SceneHolder(s)
    pv = pvol.PhotonVolume(p)
    pv.set_scene(h)
    assert pv.accel_info()[0] == n_tris
    pv.upload_photons(*load_photons("mesh"))
    xres, yres, spp, n_tasks = 32, 20, 2, 4
    tasks = np.arange(n_tasks, dtype=np.uint32)
    cam = abi.perspective_camera(float(s["camera.fov"][0]), xres, yres, s["camera.c2w"])
    film = abi.make_film(xres, yres, orc.gaussian_filter_table())
    smp = abi.make_sampler(xres, yres, spp, n_tasks)
    o = orc.Oracle(h, p)
    o.set_photons(*load_photons("mesh"))
    ref = orc.render_tasks(o, cam, film, smp, tasks, n_threads=4)
    r = _render(torch, pv, cam, film, smp, tasks, len(ref["rays"]))
    assert (r["rays"]["maxt"] == ref["rays"]["maxt"]).all()        # closest hit, bit for bit
    assert (r["streams"]["end_draw"] == ref["end_draws"]).all()     # every shadow-ray outcome of every march step (draw counts)
    pv.close()
