"""Shared test plumbing.  `-m "not gpu"` covers the oracle against the golden vectors, host
logic and the C-ABI surface; `-m gpu` holds the HIP-vs-oracle parity tests proper."""
import importlib
import os
import sys

import numpy as np
import pytest

try:   # torch ships its own HIP runtime: load it BEFORE libpvol.so pulls in /opt/rocm's, or torch finds no GPU afterwards
    import torch  # noqa: F401
except Exception:   # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pkg = importlib.import_module("cs348b-pbrt_amd")
abi, blob = pkg.abi, pkg.blob


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: a CPU test of a minute or more (still part of -m 'not gpu')")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.build()
    return _orc


def load_scene(name):
    return blob.load(os.path.join(GOLD, "scene_%s.bin" % name))


def load_photons(tag):
    b = blob.load(os.path.join(GOLD, "photons_%s.bin" % tag))
    return b["p"].reshape(-1, 3), b["wi"].reshape(-1, 3), b["alpha"].reshape(-1, 30)


# golden Li() cases: name -> (scene, photon map tag)
LI_CASES = {
    "vh": ("volumescene_h", "vh"),
    "vh_sparse": ("volumescene_h", "vh"),
    "vh_k500": ("volumescene_h", "vh"),
    "vh_nomap": ("volumescene_h", None),
    "rainbow": ("volumescene_rainbow", None),
    "grid16": ("volumescene_grid16", "grid16"),
    "pf": ("pinkfloyd", "pf"),
    "pf_k50": ("pinkfloyd", "pf"),
    "vhg": ("volumescene_hg", "vhg"),        # Henyey-Greenstein g = 0.6 (row a16): phase_hg in L_d and in the flux sum (wi4 reads)
    "vhg_k20": ("volumescene_hg", "vhg"),
    "mesh": ("meshroom", "mesh"),            # 966 triangles (row f4): the reference's BVHAccel / the device-built hierarchy
    "sph": ("sphereroom", "sph"),            # Shape "sphere" (row f3): a glass ball and a partial matte sphere, spot + point light
}
TRANS_CASES = {"trans_vh": "volumescene_h", "trans_grid16": "volumescene_grid16"}


def load_li_case(name):
    """Returns (scene blob, params, rays, streams, case blob) for a golden Li()/Transmittance() case."""
    c = blob.load(os.path.join(GOLD, "li_%s.bin" % name))
    scene_name = LI_CASES[name][0] if name in LI_CASES else TRANS_CASES[name]
    s = load_scene(scene_name)
    p = abi.params_from_blob(s, step_size=float(c["params.f"][0]), max_dist=float(c["params.f"][1]),
                             n_used=int(c["params.nused"][0]))
    rays = abi.make_rays(c["rays.o"].reshape(-1, 3), c["rays.d"].reshape(-1, 3), c["rays.mint"], c["rays.maxt"],
                         c["rays.u"], c["rays.time"], c["rays.skip"])
    streams = abi.make_streams(c["streams.seed"], c["streams.n"], c["streams.start"])
    return s, p, rays, streams, c


# golden whole-task renders (ref_capture `render`): name -> (scene, photon map tag)
RENDER_CASES = {
    "vh": ("volumescene_h", "vh"),          # 1 distant light, homogeneous: COUNT pre-pass + li_par_kernel
    "vh64": ("volumescene_h", "vh"),        # 64 spp: one pixel per wave in the film kernel
    "grid16": ("volumescene_grid16", "grid16"),   # VolumeGrid: fused RESOLVE pre-pass + replay
    "pf": ("pinkfloyd", "pf"),              # spot light through a glass prism's triangles
    "mesh": ("meshroom", "mesh"),           # camera rays and shadow rays against a 960-triangle ball (row f4)
    "sph": ("sphereroom", "sph"),           # camera rays clipped by spheres, two lights (FUSED pre-pass) (row f3)
}
# the same with the reference's surface integrator in place (ref_capture `render ... surface`): scene, photon map tag
RENDER_SURF_CASES = {"vh_surf": ("volumescene_h", "vh"), "vh_surf64": ("volumescene_h", "vh")}
# ... and with specular surfaces in view: the recursion of SpecularReflect / SpecularTransmit (oracle only so far; the device refuses)
RENDER_SPECULAR_CASES = {"pf_surf": ("pinkfloyd", "pf"), "sph_surf": ("sphereroom", "sph")}


def load_render_case(name):
    """Returns (scene blob, params, camera, film, sampler, case blob) for a golden render case."""
    c = blob.load(os.path.join(GOLD, "render_%s.bin" % name))
    s = load_scene((RENDER_CASES[name] if name in RENDER_CASES else RENDER_SURF_CASES[name] if name in RENDER_SURF_CASES else RENDER_SPECULAR_CASES[name])[0])
    p = abi.params_from_blob(s, step_size=float(c["params.f"][0]), max_dist=float(c["params.f"][1]),
                             n_used=int(c["params.nused"][0]))
    si = c["sampler.i"]
    cam = abi.make_camera(c["camera.raster_to_camera"], c["camera.camera_to_world"], float(c["camera.shutter_lens"][0]),
                          float(c["camera.shutter_lens"][1]), float(c["camera.shutter_lens"][2]), float(c["camera.shutter_lens"][3]))
    film = abi.make_film(int(si[0]), int(si[1]), c["film.filter_table"], float(c["film.filter_width"][0]), float(c["film.filter_width"][1]))
    smp = abi.make_sampler(int(si[0]), int(si[1]), int(si[2]), int(si[3]), float(c["film.filter_width"][0]), float(c["film.filter_width"][1]),
                           n1d=tuple(int(v) for v in c["sampler.n1d"]), n2d=tuple(int(v) for v in c["sampler.n2d"]),
                           tau_index=int(si[4]), scatter_index=int(si[5]))
    return s, p, cam, film, smp, c


def rel_l2(a, b, axis=-1, floor=1e-30):
    """Per-row relative L2 error ||a-b|| / max(||b||, floor)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b, axis=axis) / np.maximum(np.linalg.norm(b, axis=axis), floor)
