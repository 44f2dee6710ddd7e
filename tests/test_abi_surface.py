"""CPU-side checks of the drop-in boundary: libpvol.so loads, exports every symbol include/pvol.h
declares, the ctypes mirrors have the C struct sizes, and nothing computes without a GPU."""
import ctypes as C
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, abi, load_scene


@pytest.fixture(scope="module")
def pvol():
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "cs348b-pbrt_amd", "csrc")])
    return importlib.import_module("cs348b-pbrt_amd.pvol")


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pvol.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pvol_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pvol):
    L = pvol.lib()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "libpvol.so does not export %s" % n
    assert sorted(pvol.EXPORTS) == names
    assert L.pvol_abi_version() == 3   # 2: pvol_render_debug.d_surf_xyz, pvol_set_surface_integrator; 3: pvol_scene.spheres, pvol_get_accel_info


def test_struct_sizes_match_the_header(pvol, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "pvol.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(pvol_ray),sizeof(pvol_stream),sizeof(pvol_params),sizeof(pvol_scene),sizeof(pvol_volume),"
                   "sizeof(pvol_light),sizeof(pvol_material),sizeof(pvol_triangle),sizeof(pvol_stats),"
                   "sizeof(pvol_camera),sizeof(pvol_film),sizeof(pvol_sampler),sizeof(pvol_render_debug),sizeof(pvol_sphere));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mirrors = [abi.RAY_DTYPE.itemsize, abi.STREAM_DTYPE.itemsize, C.sizeof(abi.Params), C.sizeof(abi.Scene), C.sizeof(abi.Volume),
               C.sizeof(abi.Light), C.sizeof(abi.Material), C.sizeof(abi.Triangle), C.sizeof(abi.Stats),
               C.sizeof(abi.Camera), C.sizeof(abi.Film), C.sizeof(abi.Sampler), C.sizeof(abi.RenderDebug), C.sizeof(abi.Sphere)]
    assert sizes == mirrors


def test_defaults_are_the_reference_defaults(pvol):
    p = abi.Params()
    pvol.lib().pvol_default_params(C.byref(p))
    # CreatePhotonVolumeIntegrator photonvolume.cpp:224-229, CreatePhotonShooter photonshooter.cpp:529-548
    assert (p.step_size, p.n_used, p.n_volume_photons, p.max_photon_depth) == (1.0, 250, 0, 5)
    assert abs(p.max_dist - 0.1) < 1e-7 and abs(p.shooter_step_size - 0.1) < 1e-7
    assert (p.n_caustic_photons, p.n_indirect_photons, p.final_gather) == (20000, 10000, 1)


def test_fails_loudly_without_a_gpu(pvol):
    """No CPU fallback: without a HIP device creation reports PVOL_E_NO_DEVICE."""
    L = pvol.lib()
    if L.pvol_device_count() > 0:
        pytest.skip("a HIP device is present")
    s = load_scene("volumescene_h")
    with pytest.raises(pvol.PvolError) as e:
        pvol.PhotonVolume(abi.params_from_blob(s))
    assert e.value.status == abi.PVOL_E_NO_DEVICE
    assert b"no CPU path" in L.pvol_strerror(abi.PVOL_E_NO_DEVICE)


def test_invalid_arguments_are_rejected(pvol):
    L = pvol.lib()
    h = C.c_void_p()
    p = abi.Params()
    L.pvol_default_params(C.byref(p))
    p.n_used = 0
    assert L.pvol_create(C.byref(p), C.byref(h)) == abi.PVOL_E_INVALID
    assert L.pvol_create(None, C.byref(h)) == abi.PVOL_E_INVALID
    assert L.pvol_set_scene(None, None) == abi.PVOL_E_INVALID
    n = C.c_uint32()
    assert L.pvol_photon_count(None, C.byref(n)) == abi.PVOL_E_INVALID


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may reference it."""
    bad = []
    for base in [os.path.join(ROOT, "cs348b-pbrt_amd"), os.path.join(ROOT, "include")]:
        for dp, _, fs in os.walk(base):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"\boracle/|liborc|\borc_|import orc", t):
                        bad.append(os.path.join(dp, f))
    assert bad == [], bad


def test_host_helpers_of_the_tile_driver_match_the_reference_capture(pvol):
    """pvol_gaussian_filter_table / pvol_compute_sub_window / pvol_render_sample_count need no GPU: checked against the
    values the reference's ImageFilm, GaussianFilter and Sampler::ComputeSubWindow produced (tests/golden/render_vh.bin)."""
    import numpy as np
    from conftest import load_render_case
    s, p, cam, film, smp, c = load_render_case("vh")
    np.testing.assert_array_equal(pvol.gaussian_filter_table(2.0, 2.0, 2.0), c["film.filter_table"])
    for i, t in enumerate(c["tasks"]):
        assert pvol.sub_window(smp, int(t)) == list(c["task.window"][4 * i:4 * i + 4])
    assert pvol.render_sample_count(smp, c["tasks"]) == int(c["task.n_samples"].sum()) == len(c["samples.time"])
