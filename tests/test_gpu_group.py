"""li_group_kernel (one camera ray per lane, 64 gathers sharing an LDS-staged photon bucket) against the oracle and
against li_par_kernel (one wave per ray) on photon maps dense enough that the bucket plan -- histogram selection of
each lane's exact k-th distance, scalar-cache flux rows -- carries the lookups (the golden Li() cases use 6 k-photon
maps whose lookups mostly run at the full radius).  Bars: radiance <= 1e-4 rel. L2 per ray against the oracle,
RNG draw counts exact, the two GPU kernels within 2e-5 of each other."""
import importlib
import os

import numpy as np
import pytest

from conftest import ROOT, abi, load_scene, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def pvol():
    m = importlib.import_module("cs348b-pbrt_amd.pvol")
    assert m.lib().pvol_device_count() >= 1
    return m


def _camera_batch(orc, scene, xres, yres, spp, n_tasks, tasks):
    """Real camera samples of whole render tasks (LD sampler + perspective camera + clip) from the oracle's tile driver."""
    cam = abi.perspective_camera(float(scene["camera.fov"][0]), xres, yres, scene["camera.c2w"])
    film = abi.make_film(xres, yres, orc.gaussian_filter_table())
    smp = abi.make_sampler(xres, yres, spp, n_tasks)
    o = orc.Oracle(abi.SceneHolder(scene), abi.params_from_blob(scene))   # no photon map: only the rays are wanted
    r = orc.render_tasks(o, cam, film, smp, np.asarray(tasks, np.uint32))
    counts = [orc.sub_window(smp, t) for t in tasks]
    counts = np.array([(w[1] - w[0]) * (w[3] - w[2]) * spp for w in counts], np.uint32)
    return r["rays"], abi.make_streams(np.asarray(tasks, np.uint32), counts)


def _dense_map(pvol, scene, n_photons, n_tasks, **over):
    p = abi.params_from_blob(scene, n_volume_photons=n_photons, **over)
    pv = pvol.PhotonVolume(p)
    pv.set_scene(abi.SceneHolder(scene))
    pv.preprocess(n_tasks)   # many virtual tasks: the device shooter runs one task per lane
    return pv, p, pv.download_photons()


@pytest.fixture(scope="module")
def vh_map(pvol):
    """One 150 k-photon map of the volumescene shot on the device, shared by the tests of this module."""
    s = load_scene("volumescene_h")
    pv, p, photons = _dense_map(pvol, s, 150000, 8192)
    pv.close()
    return s, p, photons


def _ctx(pvol, s, p, photons):
    pv = pvol.PhotonVolume(p)
    pv.set_scene(abi.SceneHolder(s))
    pv.upload_photons(*photons)
    return pv


@pytest.mark.parametrize("scene_name,n_photons,over", [("volumescene_h", 150000, {}),
                                                       ("shootbench", 300000, {"n_used": 50, "max_dist": 0.5, "step_size": 0.2}),
                                                       ("meshroom", 150000, {})])   # shadow rays through the device-built triangle hierarchy (row f4)
def test_group_kernel_matches_oracle_on_a_dense_map(pvol, orc, vh_map, scene_name, n_photons, over):
    """volumescene: distant light; shootbench: a SPOT light through a glass prism's triangles (falloff, 1/d^2, occlusion);
    meshroom: 966 triangles, every shadow ray of the kernel walks the LBVH."""
    if scene_name == "volumescene_h":
        s, p, photons = vh_map
        pv = _ctx(pvol, s, p, photons)
    else:
        # the device shooter needs ~100 s for this prism scene whatever the photon count (whole 4096-path blocks per task):
        # a seeded uniform map inside the medium serves the gather parity just as well
        s = load_scene(scene_name)
        p = abi.params_from_blob(s, **over)
        rng = np.random.default_rng(11)
        ext = s["vol.extent"].astype(np.float64)
        v2w = s["vol.v2w"].reshape(4, 4).astype(np.float64)
        pvl = ext[:3] + (ext[3:] - ext[:3]) * rng.random((n_photons, 3))
        P = (pvl @ v2w[:3, :3].T + v2w[:3, 3]).astype(np.float32)
        W = rng.normal(size=(n_photons, 3)).astype(np.float32)
        W /= np.linalg.norm(W, axis=1, keepdims=True)
        A = (rng.random((n_photons, 30)) * 1e-3).astype(np.float32)
        photons = (P, W, A)
        pv = _ctx(pvol, s, p, photons)
    try:
        assert pv.photon_count() >= n_photons
        rays, streams = _camera_batch(orc, s, 640, 360, 64, 4096, [700, 2100])   # ~55 pixels x 64 spp per task
        assert len(rays) >= 4096
        pv.enable_stats(True)
        pv.stats(reset=True)
        got, gd = pv.li(rays, streams.copy())
        assert pv.march_kernel_name() == "li_group_kernel"
        st = pv.stats()
        o = orc.Oracle(abi.SceneHolder(s), p)
        o.set_photons(*photons)
        ref, rd = o.li_batch(rays, streams.copy(), n_threads=8)
        assert (gd == rd).all()
        lit = np.linalg.norm(ref[:, :30], axis=1) > 0
        assert lit.sum() > len(rays) // 4
        err = rel_l2(got[:, :30], ref[:, :30], floor=1e-12)
        assert err.max() <= TOL, "rel L2 %.3g at ray %d" % (err.max(), int(err.argmax()))
        np.testing.assert_allclose(got[:, 30:], ref[:, 30:], rtol=1e-5, atol=1e-7)
        # the bucket plan, not the exact lookup, served the lookups (a 7 k-ray batch is 14 chunks = 14 waves, each of which
        # starts cold; the headline-shape test below holds the 5 % bar on whole render tasks)
        assert st["n_guess_retries"] <= 0.10 * st["n_steps"], st
        assert st["n_kept"] >= 0.8 * p.n_used * (st["n_steps"] - st["n_guess_retries"] - st["n_lookups_lt10"]), st
        # one stream position per render task, all draws accounted for
        o_end = streams.copy()
        o.li_batch(rays, o_end)
        g_end = streams.copy()
        pv.li(rays, g_end)
        assert (g_end["end_draw"] == o_end["end_draw"]).all()
    finally:
        pv.close()


def test_group_and_per_ray_kernels_agree(pvol, orc, vh_map, monkeypatch):
    s, p, photons = vh_map
    pv = _ctx(pvol, s, p, photons)
    monkeypatch.setenv("PVOL_NO_GROUP", "1")
    pv1 = pvol.PhotonVolume(p)
    monkeypatch.delenv("PVOL_NO_GROUP")
    try:
        pv1.set_scene(abi.SceneHolder(s))
        pv1.upload_photons(*photons)
        rays, streams = _camera_batch(orc, s, 640, 360, 64, 2048, list(range(40, 2048, 400)))
        assert len(rays) > 15000
        a, da = pv.li(rays, streams.copy())
        b, db = pv1.li(rays, streams.copy())
        assert pv.march_kernel_name() == "li_group_kernel" and pv1.march_kernel_name() == "li_par_kernel"
        assert (da == db).all()
        scale = np.abs(b[:, :30]).max()
        np.testing.assert_allclose(a[:, :30], b[:, :30], rtol=2e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(a[:, 30:], b[:, 30:], rtol=2e-6)
        a2, _ = pv.li(rays, streams.copy())
        # wave scheduling changes the guessed radii (and with them which lookups go to the exact-lookup pass), never the k-NN
        # sets: a repeat differs by the order of fp32 additions only
        np.testing.assert_allclose(a, a2, rtol=5e-6, atol=1e-7 * scale)
    finally:
        pv.close()
        pv1.close()


def test_ragged_batches_and_rays_that_miss(pvol, orc, vh_map):
    """Batch sizes around the 64-lane / 256-ray chunk boundaries, rays that miss the volume, several streams per chunk."""
    s, p, photons = vh_map
    pv = _ctx(pvol, s, p, photons)
    o = orc.Oracle(abi.SceneHolder(s), p)
    o.set_photons(*photons)
    try:
        rays, _ = _camera_batch(orc, s, 640, 360, 16, 4096, [1000])
        miss = rays[:7].copy()
        miss["d"] = -miss["d"]          # away from the medium
        rays = np.concatenate([miss, rays])
        for n in (1, 63, 65, 256, 257, 300):
            sub = rays[:n].copy()
            a, b2 = n // 3, n // 3
            counts = np.array([a, b2, n - a - b2], np.uint32)
            st = abi.make_streams(np.array([11, 12, 13], np.uint32), counts)
            got, gd = pv.li(sub, st.copy())
            ref, rd = o.li_batch(sub, st.copy())
            assert (gd == rd).all()
            assert rel_l2(got[:, :30], ref[:, :30], floor=1e-12).max() <= TOL
            np.testing.assert_allclose(got[:, 30:], ref[:, 30:], rtol=1e-5, atol=1e-7)
    finally:
        pv.close()


def test_headline_shape_256spp_on_a_million_photon_map(pvol, orc):
    """The configuration bench.py quotes (C2: 1280x720, 256 spp, 4096 render tasks, >= 1 M photons shot on the device with
    16384 virtual tasks) through the entry point it uses (pvol_render_tasks_device: tile pre-pass, li_group_kernel with its
    512-ray chunks = two pixels at 256 spp, film), for a handful of WHOLE render tasks against the oracle's
    SamplerRendererTask loop on the same downloaded map: per-sample XYZ <= 1e-4 rel. L2, every stream position exact."""
    import torch
    s = load_scene("volumescene_h")
    pv, p, photons = _dense_map(pvol, s, 1000000, 16384)
    try:
        assert pv.photon_count() >= 1000000
        xres, yres, spp, ntasks = 1280, 720, 256, 4096
        cam = abi.perspective_camera(float(s["camera.fov"][0]), xres, yres, s["camera.c2w"])
        film = abi.make_film(xres, yres, pvol.gaussian_filter_table())
        smp = abi.make_sampler(xres, yres, spp, ntasks)
        tasks = np.array([37, 1500, 2050, 4000], np.uint32)   # frame corner (apron, rays that miss), middle rows, last row
        n = pvol.render_sample_count(smp, tasks)
        assert n > 200000
        dev = torch.device("cuda:0")
        pixels = torch.zeros((yres, xres, 4), dtype=torch.float32, device=dev)
        xyz = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        streams = torch.zeros((len(tasks), 32), dtype=torch.uint8, device=dev)
        pv.enable_stats(True)
        pv.stats(reset=True)
        pv.render_tasks(cam, film, smp, tasks, pixels.data_ptr(), abi.RenderDebug(0, 0, xyz.data_ptr(), streams.data_ptr()))
        torch.cuda.synchronize()
        pv.check_errors()
        assert pv.march_kernel_name() == "li_group_kernel"
        st = pv.stats()
        o = orc.Oracle(abi.SceneHolder(s), p)
        o.set_photons(*photons)
        ref = orc.render_tasks(o, cam, film, smp, tasks, n_threads=8)
        end = streams.cpu().numpy().view(abi.STREAM_DTYPE).reshape(-1)["end_draw"]
        np.testing.assert_array_equal(end, ref["end_draws"])
        a, b = xyz.cpu().numpy().astype(np.float64), ref["xyzT"].astype(np.float64)
        scale = np.abs(b[:, :3]).max()
        err = np.linalg.norm(a[:, :3] - b[:, :3], axis=1) / np.maximum(np.linalg.norm(b[:, :3], axis=1), 1e-6 * scale)
        # the north_star's bar: <= 1e-4 relative L2 PER PIXEL (a pixel = the mean of its 256 samples)
        ap, bp = a[:, :3].reshape(-1, spp, 3).mean(1), b[:, :3].reshape(-1, spp, 3).mean(1)
        perr = np.linalg.norm(ap - bp, axis=1) / np.maximum(np.linalg.norm(bp, axis=1), 1e-6 * scale)
        assert perr.max() <= TOL, "per-pixel XYZ rel L2 %.3g at pixel %d" % (perr.max(), int(perr.argmax()))
        # per SAMPLE the same bar holds except where two photons tie EXACTLY (same fp32 DistanceSquared) for the k-th
        # place of one of the sample's ~38 lookups: the reference keeps whichever its kd-tree traversal met first
        # (kdtree.h:180 rejects `dist2 == maxDistSquared`), an order that exists only inside std::nth_element's layout;
        # the grid keeps the first in cell order.  Either set is a valid k-NN set with the same radius; the two differ by
        # one photon of 50 in one step of ~38 (<= ~5e-4 of the sample).  Expected ~7e-6 ties per lookup = 3e-4 per sample.
        assert (err > TOL).mean() <= 1e-3, "%d samples above 1e-4" % int((err > TOL).sum())
        assert err.max() <= 2e-3, "per-sample XYZ rel L2 %.3g at sample %d" % (err.max(), int(err.argmax()))
        np.testing.assert_allclose(a[:, 3], b[:, 3], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(pixels.cpu().numpy(), ref["pixels"], rtol=1e-4, atol=1e-5 * np.abs(ref["pixels"]).max())
        assert st["n_guess_retries"] <= 0.05 * st["n_steps"], st   # the bucket plan, not the exact lookup, is what ran
    finally:
        pv.close()


def _grid_scene(n):
    """volumescene with the synthetic config-4 VolumeGrid at n^3 (the generator of tools/measure_configs.py)."""
    s = dict(load_scene("volumescene_grid16"))
    s["vol.dims"] = np.array([n, n, n], np.int32)
    g = (np.arange(n) + .5) / n
    zz, yy, xx = np.meshgrid(g, g, g, indexing="ij")
    rng = np.random.default_rng(348)
    dens = 0.5 + 0.5 * np.sin(7 * xx) * np.sin(5 * yy) * np.sin(3 * zz) + 0.25 * (rng.random((n, n, n)) - .5)
    s["vol.density"] = np.clip(dens, 0, 1.5).astype(np.float32).reshape(-1)
    return s


@pytest.mark.parametrize("case", ["grid128", "pinkfloyd_k500", "pinkfloyd_k500_exactpass", "pinkfloyd_k50",
                                  "pinkfloyd_k500_subgrid0", "pinkfloyd_k500_subgrid1", "vh_subgrid0", "vh_subgrid1"])
def test_group_replay_form_matches_oracle(pvol, orc, case):
    """Scenes where drawn values reach the result run as RNG pre-pass + li_group_kernel's REPLAY form (+ the exact-lookup pass):
    C4's 128^3 VolumeGrid (trilinear density and stepped tau() per lane, recorded offsets), and pinkfloyd's two lights with
    C3's nused 500 (fixed-radius plan, dense lookups handed to li_fixup_group_kernel: 64 lookups per staged bucket, histogram
    selection of the 500 nearest; `_exactpass`: to li_fixup_kernel, one wave per lookup) and with nused 50 (the bucket plan proper)."""
    import os
    old_env = os.environ.pop("PVOL_FIX_EXACT", None)
    old_sub = os.environ.pop("PVOL_SUBGRID", None)
    if case == "pinkfloyd_k500_exactpass":
        os.environ["PVOL_FIX_EXACT"] = "1"
    # the second grid level (4 x 4 x 4 sub-cells, chosen by a data-dependent threshold when the map is built) forced off and on:
    # results do not depend on it (`_subgridN`; the variable is read when the map is finished, i.e. inside preprocess)
    if "_subgrid" in case:
        os.environ["PVOL_SUBGRID"] = case[-1]
    if case.startswith("vh_"):
        s = load_scene("volumescene_h")
        over, n_photons, n_tasks, res, spp, tasks = {}, 150000, 8192, (256, 256), 16, [40, 130, 200]
    elif case == "grid128":
        s = _grid_scene(128)
        over, n_photons, n_tasks, res, spp, tasks = {}, 200000, 2048, (256, 256), 16, [40, 130, 200]
    elif case.startswith("pinkfloyd_k500"):
        s = load_scene("pinkfloyd")
        over, n_photons, n_tasks, res, spp, tasks = {"n_caustic_photons": 0}, 400000, 64, (480, 270), 4, [100, 230, 300, 410]
    else:
        s = load_scene("pinkfloyd")
        over, n_photons, n_tasks, res, spp, tasks = {"n_caustic_photons": 0, "n_used": 50, "max_dist": 0.25}, 400000, 64, (480, 270), 4, [100, 230, 300, 410]
    pv, p, photons = _dense_map(pvol, s, n_photons, n_tasks, **over)
    try:
        assert pv.photon_count() >= n_photons
        rays, streams = _camera_batch(orc, s, res[0], res[1], spp, 512, tasks)
        assert len(rays) >= 2000
        got, gd = pv.li(rays, streams.copy())
        assert pv.march_kernel_name() == "li_group_kernel"
        pv.check_errors()
        o = orc.Oracle(abi.SceneHolder(s), p)
        o.set_photons(*photons)
        ref, rd = o.li_batch(rays, streams.copy(), n_threads=8)
        assert (gd == rd).all()
        lit = np.linalg.norm(ref[:, :30], axis=1) > 0
        assert lit.sum() > len(rays) // 8
        floor = 1e-6 * float(np.abs(ref[:, :30]).max())
        err = rel_l2(got[:, :30], ref[:, :30], floor=floor)
        # exact ties for the k-th place aside (see the headline test), every ray is within the bar
        assert (err > TOL).mean() <= 2e-3 and err.max() <= 2e-3, "rel L2 %.3g at ray %d, %d above 1e-4" % (err.max(), int(err.argmax()), int((err > TOL).sum()))
        np.testing.assert_allclose(got[:, 30:], ref[:, 30:], rtol=1e-5, atol=1e-7)
        g_end, o_end = streams.copy(), streams.copy()
        pv.li(rays, g_end)
        o.li_batch(rays, o_end)
        assert (g_end["end_draw"] == o_end["end_draw"]).all()
        if case == "pinkfloyd_k500":   # the shared-bucket pass did the work, the exact pass the odd lookup
            st = pv.stats()
            assert st["group_plan_skipped"] > 20 * max(1, st["group_guess_failed"]), (st["group_plan_skipped"], st["group_guess_failed"])
    finally:
        pv.close()
        os.environ.pop("PVOL_FIX_EXACT", None)
        os.environ.pop("PVOL_SUBGRID", None)
        if old_env is not None:
            os.environ["PVOL_FIX_EXACT"] = old_env
        if old_sub is not None:
            os.environ["PVOL_SUBGRID"] = old_sub
