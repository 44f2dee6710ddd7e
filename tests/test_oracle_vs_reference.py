"""Pins the CPU oracle (oracle/) to golden vectors written by the REFERENCE's own objects
(oracle/ref_capture.cpp linked against oracle/_ref/libpbrtref.a; tests/golden/ref_*.bin and the
`ref.*` entries of tests/golden/li_*.bin).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import (GOLD, LI_CASES, RENDER_CASES, TRANS_CASES, abi, blob, load_li_case, load_photons, load_render_case,
                      load_scene, rel_l2)

F = C.POINTER(C.c_float)
U = C.POINTER(C.c_uint32)


def fp(a):
    return a.ctypes.data_as(F)


@pytest.fixture(scope="module")
def tables():
    return blob.load(os.path.join(GOLD, "ref_tables.bin"))


def test_rng_matches_reference(orc, tables):
    # core/rng.cpp:43-107; 1300 draws cross two table regenerations
    L = orc.lib()
    out = np.zeros(1300, np.uint32)
    fl = np.zeros(64, np.float32)
    for i, s in enumerate(tables["rng.seeds"]):
        L.orc_rng_draws(int(s), 1300, out.ctypes.data_as(U))
        assert (out == tables["rng.draws"][1300 * i:1300 * (i + 1)]).all()
        L.orc_rng_floats(int(s), 64, fp(fl))
        assert (fl == tables["rng.floats"][64 * i:64 * (i + 1)]).all()
    # known-answer values recorded in SURVEY 8(c) from the compiled reference
    L.orc_rng_draws(0, 3, out.ctypes.data_as(U))
    assert list(out[:3]) == [2357136044, 2546248239, 3071714933]


def test_permuted_halton_matches_reference(orc, tables):
    # core/montecarlo.h:206-243, montecarlo.cpp:380-397 with RNG(31*task) (photonshooter.cpp:235,243)
    L = orc.lib()
    idx = tables["halton.index"]
    ref = tables["halton.samples"].reshape(3, len(idx), 6)
    o = np.zeros(6, np.float32)
    for t in range(3):
        for k, i in enumerate(idx):
            L.orc_halton(31 * t, 6, int(i), 1, fp(o))
            assert (o == ref[t, k]).all(), (t, i)
    L.orc_halton(0, 6, 1, 1, fp(o))
    np.testing.assert_allclose(o, [0.5, 0.33333334, 0, 0.42857143, 0.45454547, 0.07692308], rtol=0, atol=1e-8)


def test_ld_shuffles_match_reference(orc, tables):
    # LDShuffleScrambled1D/2D as Li() uses them (nSamples=1): photonvolume.cpp:137-142
    L = orc.lib()
    off = 0
    for k, n in enumerate(tables["ld.n"]):
        n = int(n)
        a = np.zeros(n, np.float32)
        draws = L.orc_ld_shuffle_1d(1000 + k, 1, n, fp(a))
        assert draws == 1 + 2 * n
        assert (a == tables["ld.1d"][off:off + n]).all()
        off += n
    a = np.zeros(64, np.float32)
    assert L.orc_ld_shuffle_1d(77, 4, 16, fp(a)) == 1 + 16 * 4 + 16
    assert (a == tables["ld.1d_4x16"]).all()


def test_sampling_routines_match_reference(orc, tables):
    L = orc.lib()
    us = tables["mc.u"].reshape(-1, 2)
    o = np.zeros(11, np.float32)
    sph, cone = tables["mc.sphere"].reshape(-1, 3), tables["mc.cone95"].reshape(-1, 3)
    disk, cosh = tables["mc.disk"].reshape(-1, 2), tables["mc.coshemi"].reshape(-1, 3)
    for i, (u1, u2) in enumerate(us):
        L.orc_mc_samples(float(u1), float(u2), fp(o))
        assert (o[0:3] == sph[i]).all() and (o[3:6] == cone[i]).all()
        assert (o[6:8] == disk[i]).all() and (o[8:11] == cosh[i]).all()
    ph = tables["phase"].reshape(-1, 5)
    o4 = np.zeros(4, np.float32)
    for row in ph:
        L.orc_phase(float(row[0]), fp(o4))
        assert (o4 == row[1:]).all()


@pytest.mark.parametrize("scene_name", ["volumescene_h", "volumescene_hg", "volumescene_rainbow", "volumescene_grid16", "pinkfloyd", "shootbench",
                                        "meshroom", "sphereroom"])
def test_scene_units_match_reference(orc, scene_name):
    """Lights, closest/any hit, BSDF sampling and volume queries: the shooter's building blocks."""
    s = load_scene(scene_name)
    u = blob.load(os.path.join(GOLD, "ref_units_%s.bin" % scene_name))
    h = abi.SceneHolder(s)
    o = orc.Oracle(h, abi.params_from_blob(s))
    L = orc.lib()
    n_l = len(s["lights.kind"])
    pw = np.zeros(n_l, np.float32)
    L.orc_light_powers(o._h, fp(pw))
    assert (pw == u["light.power_y"]).all()
    # cie / luminance
    assert np.float32(L.orc_spec_y(o._h, fp(np.ones(30, np.float32)))) == np.float32(0.99941075)
    # emission (Light::Sample_L(scene, ...))
    ein, eout = u["emit.in"].reshape(-1, 3), u["emit.out"].reshape(-1, 40)
    rec = np.zeros(40, np.float32)
    for a, b in zip(ein, eout):
        L.orc_light_emit(o._h, int(a[0]), float(a[1]), float(a[2]), fp(rec))
        assert (rec == b).all()
    # Light::Sample_L(p, ...)
    sin_, sout = u["sample.in"].reshape(-1, 4), u["sample.out"].reshape(-1, 42)
    rec = np.zeros(42, np.float32)
    for a, b in zip(sin_, sout):
        L.orc_light_sample(o._h, int(a[0]), fp(np.ascontiguousarray(a[1:4])), fp(rec))
        np.testing.assert_array_equal(rec, b)
    # closest hit, any hit, BSDF::Sample_f
    hin, hout = u["hit.in"].reshape(-1, 7), u["hit.out"].reshape(-1, 12)
    bs = u["bsdf.out"].reshape(-1, 2, 41)
    hrec = np.zeros(11, np.float32)
    brec = np.zeros(35, np.float32)
    n_hits = 0
    for a, b, bb in zip(hin, hout, bs):
        oo, dd, maxt = np.ascontiguousarray(a[0:3]), np.ascontiguousarray(a[3:6]), float(a[6])
        hit = L.orc_intersect(o._h, fp(oo), fp(dd), 0.0, maxt, fp(hrec))
        assert hit == int(b[0])
        assert L.orc_intersect_p(o._h, fp(oo), fp(dd), 0.0, maxt) == int(b[1])
        if not hit:
            continue
        n_hits += 1
        assert (hrec[0:10] == b[2:12]).all()
        tri = int(hrec[10])
        for variant in range(2):
            r = bb[variant]
            alpha = np.ones(30, np.float32)
            if variant == 1:
                alpha = np.zeros(30, np.float32)
                alpha[int(r[3])] = 0.7
            wo = np.ascontiguousarray(-dd)
            L.orc_bsdf_sample(o._h, tri, fp(wo), fp(alpha), float(r[0]), float(r[1]), float(r[2]),
                              fp(np.ascontiguousarray(hrec[7:10])), fp(np.ascontiguousarray(hrec[4:7])), fp(brec))
            assert brec[3] == r[7], "pdf"
            assert int(brec[4]) == int(r[8]), "sampled BxDFType"
            if r[7] != 0:
                assert (brec[0:3] == r[4:7]).all(), "wi"
                assert (brec[5:35] == r[11:41]).all(), "f"
    assert n_hits > 100
    # volume: IntersectP, tau, sigma_a/s, phase
    vin, vout = u["vol.in"].reshape(-1, 9), u["vol.out"].reshape(-1, 94)
    rec = np.zeros(94, np.float32)
    for a, b in zip(vin, vout):
        L.orc_volume_query(o._h, fp(np.ascontiguousarray(a)), fp(rec))
        assert rec[0] == b[0]
        if b[0]:
            assert (rec[1:3] == b[1:3]).all()
        assert (rec[3:] == b[3:]).all()
    if "rainbow" in u:
        rb = u["rainbow"].reshape(-1, 36)
        Ld = np.ascontiguousarray(u["rainbow.Ld"])
        out = np.zeros(30, np.float32)
        for row in rb:
            L.orc_rainbow(fp(Ld), fp(np.ascontiguousarray(row[0:3])), fp(np.ascontiguousarray(row[3:6])), fp(out))
            assert (out == row[6:]).all()


@pytest.mark.parametrize("name", sorted(LI_CASES))
def test_li_matches_reference(orc, name):
    """PhotonVolumeIntegrator::Li (photonvolume.cpp:112-222) through the kd-tree gather: radiance,
    transmittance, per-ray RNG draw counts and the stream position after the batch."""
    s, p, rays, streams, c = load_li_case(name)
    o = orc.Oracle(abi.SceneHolder(s), p)
    tag = LI_CASES[name][1]
    if tag:
        o.set_photons(*load_photons(tag))
    out, draws = o.li_batch(rays, streams)
    Lr, Tr = c["ref.Lv"].reshape(-1, 30), c["ref.T"].reshape(-1, 30)
    # same compiler, same libm, same heap order: the restatement is bit-identical here
    assert rel_l2(out[:, :30], Lr).max() <= 1e-6
    assert rel_l2(out[:, 30:], Tr).max() <= 1e-6
    assert (draws == c["ref.draws"]).all()
    assert (streams["end_draw"] == c["ref.streams.end"]).all()
    # the NEXT RandomUInt of every stream pins the alignment end to end
    L = orc.lib()
    for st, nxt in zip(streams, c["ref.next_rng"]):
        n = int(st["end_draw"]) + 1
        buf = np.zeros(n, np.uint32)
        L.orc_rng_draws(int(st["seed"]), n, buf.ctypes.data_as(U))
        assert buf[-1] == nxt


@pytest.mark.parametrize("name", sorted(TRANS_CASES))
def test_transmittance_matches_reference(orc, name):
    # PhotonVolumeIntegrator::Transmittance with sample == NULL (photonvolume.cpp:15-30)
    s, p, rays, streams, c = load_li_case(name)
    o = orc.Oracle(abi.SceneHolder(s), p)
    T = o.transmittance_batch(rays, streams)
    assert rel_l2(T, c["ref.T"].reshape(-1, 30)).max() <= 1e-6
    assert (streams["end_draw"] == c["ref.streams.end"]).all()


def test_shooter_statistics_match_survey(orc):
    """followPhoton as a whole cannot be pinned against the reference here (its task system does not
    build in this image; oracle/Makefile).  Anchor it on the work counters SURVEY 6 measured from the
    compiled reference at --ncores 1 on this very scene: 432 paths per stored volume photon, 65 % of
    followPhoton calls end without a surface hit, 66.6 % of interactions are absorbed, 8.0 march
    steps per emitted path."""
    b = blob.load(os.path.join(GOLD, "photons_vh.bin"))
    st = dict(zip(orc.SHOOT_STAT_NAMES, [int(x) for x in b["shoot_stats"]]))
    assert st["stored_volume"] >= 6000
    assert 400 < st["paths"] / st["stored_volume"] < 470
    assert abs(st["no_hit"] / st["follow_calls"] - 0.65) < 0.01
    assert abs(st["absorbed"] / st["interactions"] - 0.666) < 0.005
    assert abs(st["march_steps"] / st["paths"] - 8.0) < 0.1
    # volume photons only come from paths that already met a surface or scattered once (photonshooter.cpp:98)
    p = b["p"].reshape(-1, 3)
    s = load_scene("volumescene_h")
    lo, hi = s["world"][:3], s["world"][3:]
    assert (p >= lo - 1e-3).all() and (p <= hi + 1e-3).all()


def test_shooter_work_counters_equal_the_compiled_reference(orc):
    """PhotonShootingTask::Run / followPhoton as a whole (core/photonshooter.cpp:47-357) cannot be linked in this image
    (core/parallel.cpp needs <sys/sysctl.h>; no stand-in is written), so the composite is pinned by the INTEGERS the survey
    measured from the compiled reference at --ncores 1 on projectScene/pinkfloyd.pbrt (SURVEY 6, rows "Photon-shoot work,
    pinkfloyd-like" and "Photon map composition"): every one of them depends on every RNG draw and every branch of the
    shooter (emission, closest hit, transmittance march, absorb/scatter, spectral split, dispersive Sample_f, roulette,
    block merge), and a single flipped decision anywhere changes all of them."""
    s = load_scene("pinkfloyd")
    h = abi.SceneHolder(s)
    # 20 k requested: 21 038 stored, 21 021 of them monochromatic (one non-zero spectral bin, spectrum.cpp:100-111)
    o = orc.Oracle(h, abi.params_from_blob(s, n_volume_photons=20000))
    assert o.shoot(1, 1) == 0
    P, W, A = o.get_photons()
    assert len(P) == 21038
    assert int(((A != 0).sum(axis=1) == 1).sum()) == 21021
    # 200 k requested: 19 blocks of 4096 paths
    o = orc.Oracle(h, abi.params_from_blob(s, n_volume_photons=200000))
    assert o.shoot(1, 1) == 0
    st = o.shoot_stats()
    assert st["paths"] == st["nshot"] == 77824 == 19 * 4096
    assert st["stored_volume"] == 200791 == len(o.get_photons()[0])
    assert st["stored_caustic"] == 36953          # 1 requested: the first block's worth is kept (block granularity)
    assert st["follow_calls"] == 2516360
    assert st["march_steps"] == 44109004
    assert st["split_children"] == 1318800


@pytest.mark.slow
def test_shooter_on_volumescene_is_parity_unpinned_and_guarded(orc):
    """PARITY UNPINNED.  The other scene the survey measured (C1-h: volumescene with the homogeneous medium, distant light, 5 000
    caustic photons, final gather on, 100 000 volume photons, --ncores 1) does NOT reproduce integer for integer: the survey's
    compiled reference stored 100 012 volume photons from "43.2 M" paths and "345.8 M" march steps (only the first
    figure is exact), the oracle stores 100 005 from 43 233 280 paths and 345 897 879 march steps.  The cause is unknown: the
    scene the oracle shoots equals the parser-built one bit for bit (tests/test_pbrt_scene.py), the reference's
    PhotonShootingTask cannot be linked here to bisect (core/parallel.cpp needs <sys/sysctl.h>), and the same oracle code IS
    integer-exact on pinkfloyd (spot + point light, prism) above -- what volumescene adds is the distant light's emission
    (DistantLight::Sample_L(scene, ...), pinned record by record in ref_units_volumescene_h) and the radiance-photon deposits of
    `finalgather true`.  Every C2 number therefore rests on a photon map whose generator is pinned on a different scene.
    This test guards the oracle's own integers so that drift is caught (about 100 s of single-thread work)."""
    s = load_scene("volumescene_h")
    o = orc.Oracle(abi.SceneHolder(s), abi.params_from_blob(s, n_volume_photons=100000))
    assert o.shoot(1, 1) == 0
    st = o.shoot_stats()
    assert (st["stored_volume"], st["paths"], st["march_steps"]) == (100005, 43233280, 345897879)
    assert st["stored_volume"] != 100012          # the reference's count: not reproduced, cause unknown


def test_oracle_surface_stores_are_what_it_counts(orc):
    """The kept caustic / direct / indirect photons are exactly the deposits the counters (pinned above) count, every one on a
    surface of the scene, with the arrival direction stored unit length; keeping them changes nothing about the volume map."""
    s = load_scene("pinkfloyd")
    h = abi.SceneHolder(s)
    p = abi.params_from_blob(s, n_volume_photons=20000)
    o = orc.Oracle(h, p)
    o.keep_surface_photons(True)
    assert o.shoot(1, 1) == 0
    st = o.shoot_stats()
    P, W, A, npaths = o.surface_photons(0)
    assert len(P) == st["stored_caustic"] == 36953 and npaths == 4096      # causticphotons 1: done after the first block
    np.testing.assert_allclose(np.linalg.norm(W, axis=1), 1.0, atol=1e-5)
    assert (A >= 0).all() and ((A != 0).sum(axis=1) >= 1).all()
    # the caustic photons of this scene land on the matte wall x = 5 (world) behind the prism
    assert np.abs(P[:, 0] - 5.0).max() < 1e-3
    assert len(o.surface_photons(1)[0]) == st["stored_direct"] == 0 and len(o.surface_photons(2)[0]) == st["stored_indirect"] == 0
    assert len(o.get_photons()[0]) == 21038                                # as without the stores


def test_shooter_is_deterministic_and_task_mode_differs(orc):
    s = load_scene("pinkfloyd")
    h = abi.SceneHolder(s)
    maps = []
    for n_tasks, n_threads in [(1, 1), (1, 1), (4, 1), (4, 4)]:
        o = orc.Oracle(h, abi.params_from_blob(s, n_volume_photons=3000))
        assert o.shoot(n_tasks, n_threads) == 0
        maps.append(o.get_photons())
    assert all((a == b).all() for a, b in zip(maps[0], maps[1]))          # --ncores 1 is reproducible
    assert all((a == b).all() for a, b in zip(maps[2], maps[3]))          # virtual tasks: thread count is irrelevant
    assert len(maps[2][0]) >= 3000
    # golden map was shot with the same code: regenerate and compare bit for bit
    o = orc.Oracle(h, abi.params_from_blob(s, n_volume_photons=6000))
    assert o.shoot(1, 1) == 0
    P, W, A = o.get_photons()
    gp, gw, ga = load_photons("pf")
    assert (P == gp).all() and (W == gw).all() and (A == ga).all()


# ---------------------------------------------------------------- tile driver
@pytest.mark.parametrize("name", list(RENDER_CASES))
def test_render_task_records_match_reference(orc, name):
    """LDSampler + PerspectiveCamera + Scene::Intersect + Li + ImageFilm, walked in SamplerRendererTask::Run order:
    every per-sample record, every stream position and the film are bit-identical to the reference's."""
    s, p, cam, film, smp, c = load_render_case(name)
    assert [smp.x_start, smp.x_end, smp.y_start, smp.y_end] == list(c["sampler.extent"])   # Film::GetSampleExtent
    for i, t in enumerate(c["tasks"]):
        assert orc.sub_window(smp, int(t)) == list(c["task.window"][4 * i:4 * i + 4])     # Sampler::ComputeSubWindow
    o = orc.Oracle(abi.SceneHolder(s), p)
    tag = RENDER_CASES[name][1]
    if tag:
        o.set_photons(*load_photons(tag))
    r = orc.render_tasks(o, cam, film, smp, c["tasks"])
    assert r["n_samples"] == len(c["samples.time"]) > 0
    np.testing.assert_array_equal(r["image_xy"].ravel(), c["samples.image"])
    np.testing.assert_array_equal(r["rays"]["time"], c["samples.time"])
    np.testing.assert_array_equal(r["rays"]["scatter_u"], c["samples.scatter"])
    np.testing.assert_array_equal(r["rays"]["o"].ravel(), c["rays.o"])
    np.testing.assert_array_equal(r["rays"]["d"].ravel(), c["rays.d"])
    np.testing.assert_array_equal(r["rays"]["mint"], c["rays.t"][0::2])
    np.testing.assert_array_equal(r["rays"]["maxt"], c["rays.t"][1::2])
    np.testing.assert_array_equal(r["rays"]["rng_skip"], c["rays.skip"])
    np.testing.assert_array_equal(r["end_draws"], c["task.end_draw"])
    np.testing.assert_array_equal(r["xyzT"].ravel(), c["xyzT"])
    np.testing.assert_array_equal(r["pixels"].ravel(), c["film.pixels"])
    np.testing.assert_array_equal(orc.film_resolve(film, r["pixels"]).ravel(), c["film.rgb"])
    assert np.isfinite(c["film.rgb"]).all() and c["film.rgb"].max() > 0


@pytest.mark.parametrize("name", ["vh_surf", "vh_surf64"])
def test_render_with_the_surface_integrator_matches_reference(orc, name):
    """SURVEY 8(f)-2, matte subset: whole SamplerRendererTasks with the reference's PhotonIntegrator in place (direct lighting,
    caustic estimate, and every RNG draw of the rest of its Li()): per-sample surface radiance, the composed T * Ls + Lvi, the
    draws in front of every volume Li(), the stream ends and the film are the reference's, bit for bit."""
    s, p, cam, film, smp, c = load_render_case(name)
    cb = blob.load(os.path.join(GOLD, "caustic_vh.bin"))
    o = orc.Oracle(abi.SceneHolder(s), p)
    o.set_photons(*load_photons("vh"))
    o.set_surface_integrator(int(c["surf.params.i"][0]), float(c["surf.params.f"][0]), bool(c["surf.params.i"][1]),
                             (cb["p"].reshape(-1, 3), cb["wo"].reshape(-1, 3), cb["alpha"].reshape(-1, 30)), int(cb["n_paths"][0]))
    assert int(c["surf.params.i"][2]) == int(cb["n_paths"][0])
    r = orc.render_tasks(o, cam, film, smp, c["tasks"])
    assert not r["unsupported_hits"]
    np.testing.assert_array_equal(r["rays"]["rng_skip"], c["rays.skip"])          # sampler draws + the surface integrator's
    np.testing.assert_array_equal(r["end_draws"], c["task.end_draw"])
    np.testing.assert_array_equal(r["surf_xyz"].ravel(), c["surf.xyz"])
    np.testing.assert_array_equal(r["xyzT"].ravel(), c["xyzT"])
    np.testing.assert_array_equal(r["pixels"].ravel(), c["film.pixels"])
    assert set(np.unique(c["surf.draws"])) <= {0, 150, 151}                       # 144 (2 x rho) + 6 (2 x BSDFSample) + unoccluded light
    assert (c["surf.xyz"].reshape(-1, 3).sum(1) > 0).mean() > 0.5


@pytest.mark.parametrize("name,photons", [("pf_surf", "pf"), ("sph_surf", "sph")])
def test_specular_recursion_matches_reference(orc, name, photons):
    """SURVEY 8(f)-2, the specular part: camera samples that meet the glass prism / the glass ball go through
    SpecularReflect + SpecularTransmit (core/integrator.cpp:177-262) -- 3 + 3 draws, a spawned ray through SamplerRenderer::Li
    with its own surface term and its own volume Li() (same Sample, so the same scatter offset), up to maxspeculardepth levels,
    all drawn from the tile's stream BEFORE the primary ray's volume term.  The oracle reproduces the reference's per-sample
    surface radiance, composed radiance, draw counts, stream ends and film bit for bit."""
    s, p, cam, film, smp, c = load_render_case(name)
    cb = blob.load(os.path.join(GOLD, "caustic_%s.bin" % photons))
    o = orc.Oracle(abi.SceneHolder(s), p)
    o.set_photons(*load_photons(photons))
    o.set_surface_integrator(int(c["surf.params.i"][0]), float(c["surf.params.f"][0]), bool(c["surf.params.i"][1]),
                             (cb["p"].reshape(-1, 3), cb["wo"].reshape(-1, 3), cb["alpha"].reshape(-1, 30)), int(cb["n_paths"][0]))
    r = orc.render_tasks(o, cam, film, smp, c["tasks"])
    assert not r["unsupported_hits"]
    np.testing.assert_array_equal(r["rays"]["rng_skip"], c["rays.skip"])
    np.testing.assert_array_equal(r["end_draws"], c["task.end_draw"])
    np.testing.assert_array_equal(r["surf_xyz"].ravel(), c["surf.xyz"])
    np.testing.assert_array_equal(r["xyzT"].ravel(), c["xyzT"])
    np.testing.assert_array_equal(r["pixels"].ravel(), c["film.pixels"])
    # the frame is aimed at the glass: a good number of samples carry more surface draws than any matte hit makes (151): the
    # nested volume Li() calls of the spawned rays are counted among them
    assert (c["surf.draws"] > 151).sum() > 20, int((c["surf.draws"] > 151).sum())


def test_ld_pixel_sample_and_filter_table_match_reference(orc):
    s, p, cam, film, smp, c = load_render_case("vh")
    np.testing.assert_array_equal(orc.gaussian_filter_table(2.0, 2.0, 2.0), c["film.filter_table"])
    # first pixel of task 0: the sampler's draws precede everything else in the stream
    w = list(c["task.window"][:4])
    ps, draws = orc.ld_pixel_sample(smp, w[0], w[2], seed=int(c["tasks"][0]))
    n = smp.pixel_samples
    assert draws == int(c["rays.skip"][0]) == 7 + 10 * n     # 2+2n, 2+2n, 1+2n, 2 x (1+2n)
    np.testing.assert_array_equal(ps["imageX"], c["samples.image"][0:2 * n:2])
    np.testing.assert_array_equal(ps["imageY"], c["samples.image"][1:2 * n:2])
    np.testing.assert_array_equal(ps["lensU"], c["samples.lens"][0:2 * n:2])
    np.testing.assert_array_equal(ps["tau"], c["samples.tau"][:n])
    np.testing.assert_array_equal(ps["scatter"], c["samples.scatter"][:n])
    # the fp32 camera helper used by bench/tests agrees with the reference's matrices
    pc = abi.perspective_camera(float(s["camera.fov"][0]), film.x_resolution, film.y_resolution, c["camera.camera_to_world"])
    np.testing.assert_allclose(np.array(pc.raster_to_camera), c["camera.raster_to_camera"], rtol=1e-6, atol=1e-6)
