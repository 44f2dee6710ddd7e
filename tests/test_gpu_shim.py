"""The drop-in boundary end to end: integration/hip_photonvolume.cpp -- `class HipPhotonVolumeIntegrator : public
VolumeIntegrator` over libpvol.so -- constructed INSIDE the reference's own object graph and driven through the reference's
virtual interface, beside the reference's PhotonVolumeIntegrator, by oracle/_ref/shim_drive (built here from the reference's
objects + libpvol.so, oracle/Makefile; oracle/shim_drive.cpp).

`li`: the rays of a golden case through `vi->Li(scene, renderer, ray, sample, rng, &T, arena)` with a live RNG: the binding
must return the reference's radiance (1e-4 rel. L2), leave the caller's RNG in the reference's state (next RandomUInt equal,
draw counts equal) -- and the reference leg of the same run must reproduce the committed fixture bit for bit, which ties this
binary to the captures everything else is held against.
`render`: whole SamplerRendererTasks -- the reference's LDSampler / PerspectiveCamera / ImageFilm objects -- three ways: the
reference integrator, the binding per sample, and the binding's RenderTasks (device tile driver): three films, equal to 1e-4."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, LI_CASES, RENDER_CASES, ROOT, blob, rel_l2

TOOL = os.path.join(ROOT, "oracle", "_ref", "shim_drive")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(TOOL), reason="oracle/_ref/shim_drive is not built (needs the reference tree at build time)")]


def _run(mode, scene, tag, case_file, tmp_path):
    out = str(tmp_path / "shim_out.bin")
    photons = os.path.join(GOLD, "photons_%s.bin" % tag) if tag else "-"
    r = subprocess.run([TOOL, mode, scene, photons, os.path.join(GOLD, case_file), out], timeout=600, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return blob.load(out)


@pytest.mark.parametrize("name", ["vh", "vh_sparse", "vh_k500", "vh_nomap", "rainbow", "grid16", "pf", "pf_k50", "vhg", "mesh", "sph"])
def test_binding_li_equals_the_reference_integrator_beside_it(name, tmp_path):
    scene, tag = LI_CASES[name]
    o = _run("li", scene, tag, "li_%s.bin" % name, tmp_path)
    gold = blob.load(os.path.join(GOLD, "li_%s.bin" % name))
    # the run's reference leg IS the fixture
    assert o["ref.Lv"].tobytes() == gold["ref.Lv"].tobytes() and o["ref.T"].tobytes() == gold["ref.T"].tobytes()
    np.testing.assert_array_equal(o["ref.draws"], gold["ref.draws"])
    np.testing.assert_array_equal(o["ref.next_rng"], gold["ref.next_rng"])
    # the binding: RNG handed in and out exactly, radiance within the north_star's bar
    np.testing.assert_array_equal(o["hip.draws"], o["ref.draws"])
    np.testing.assert_array_equal(o["hip.next_rng"], o["ref.next_rng"])
    ref, hip = o["ref.Lv"].reshape(-1, 30), o["hip.Lv"].reshape(-1, 30)
    scale = np.linalg.norm(ref.astype(np.float64), axis=1).max()
    assert rel_l2(hip, ref, floor=1e-3 * max(scale, 1e-30)).max() <= 1e-4
    assert rel_l2(o["hip.T"].reshape(-1, 30), o["ref.T"].reshape(-1, 30)).max() <= 1e-5


@pytest.mark.parametrize("name", ["vh", "pf", "grid16", "sph"])
def test_binding_renders_the_reference_film(name, tmp_path):
    scene, tag = RENDER_CASES[name]
    o = _run("render", scene, tag, "render_%s.bin" % name, tmp_path)
    gold = blob.load(os.path.join(GOLD, "render_%s.bin" % name))
    assert o["ref.film.pixels"].tobytes() == gold["film.pixels"].tobytes()
    np.testing.assert_array_equal(o["ref.next_rng"], gold["next_rng"])
    np.testing.assert_array_equal(o["hip.next_rng"], o["ref.next_rng"])      # every task's stream ends where the reference's does
    ref = o["ref.film.pixels"].reshape(-1, 4)
    for leg in ("hip.film.pixels", "tiles.film.pixels"):
        got = o[leg].reshape(-1, 4)
        np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=2e-6, atol=1e-7, err_msg=leg)   # filter weights
        lit = np.linalg.norm(ref[:, :3], axis=1) > 1e-6 * np.linalg.norm(ref[:, :3], axis=1).max()
        assert rel_l2(got[lit, :3], ref[lit, :3]).max() <= 1e-4, leg
    np.testing.assert_allclose(o["tiles.film.rgb"], gold["film.rgb"], rtol=2e-4, atol=1e-6)
