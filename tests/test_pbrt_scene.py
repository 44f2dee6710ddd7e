"""Scene-file front end (cs348b-pbrt_amd/pbrt_scene.py, SURVEY 8(f)-3: trianglemesh and sphere shapes) against what the reference's
own parser + API layer built: the scene blobs of tests/golden/ were written by oracle/ref_capture.cpp through the reference's
Create*() functions, ParamSet and Transform classes; the front end reproduces them bit for bit."""
import importlib
import os

import numpy as np
import pytest

from conftest import GOLD, blob, load_scene

ps = importlib.import_module("cs348b-pbrt_amd.pbrt_scene")
abi = importlib.import_module("cs348b-pbrt_amd").abi
SCENES = os.path.join(GOLD, "scenes")
REF_SCENES = "/root/reference/projectScene"   # present in the build container only


def _same(parsed, ref):
    for k, v in ref.items():
        assert k in parsed, k
        np.testing.assert_array_equal(np.asarray(parsed[k]), np.asarray(v), err_msg=k)


def test_from_rgb_matches_reference_conversions():
    t = blob.load(os.path.join(GOLD, "ref_tables.bin"))
    for i, rgb in enumerate(t["rgb.colour_in"].reshape(-1, 3)):     # one colour per branch of FromRGB, white, black
        np.testing.assert_array_equal(ps.from_rgb(rgb), t["rgb.colour_refl"].reshape(-1, 30)[i])
        np.testing.assert_array_equal(ps.from_rgb(rgb, illuminant=True), t["rgb.colour_illum"].reshape(-1, 30)[i])
    for i, g in enumerate(t["rgb.grey"]):                            # the greys the BASELINE scenes use
        np.testing.assert_array_equal(ps.from_rgb((g, g, g)), t["rgb.spectra"].reshape(-1, 30)[i])


@pytest.mark.parametrize("fixture,scene", [("volumescene_equiv.pbrt", "volumescene_rainbow"), ("pinkfloyd_equiv.pbrt", "pinkfloyd"),
                                            ("spherescene_equiv.pbrt", "spherescene")])   # Shape "sphere"
def test_fixture_scenes_equal_the_scenes_the_reference_built(fixture, scene):
    d = ps.load(os.path.join(SCENES, fixture))
    _same(d, load_scene(scene))
    # and the dictionary is what the rest of the package takes
    holder = abi.SceneHolder(d)
    assert holder.scene.n_triangles == len(d["tris.material"]) and holder.scene.n_lights == len(d["lights.kind"])
    p = abi.params_from_blob(d)
    assert p.n_used == int(d["params.i"][0]) and abs(p.step_size - float(d["params.f"][0])) < 1e-9


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("fname,scene", [("volumescene_png.pbrt", "volumescene_rainbow"), ("pinkfloyd.pbrt", "pinkfloyd"), ("scene.pbrt", "spherescene")])
def test_reference_scene_files_equal_the_scenes_the_reference_built(fname, scene):
    _same(ps.load(os.path.join(REF_SCENES, fname)), load_scene(scene))


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="the reference tree is not on this machine")
def test_every_reference_scene_file_parses_or_is_refused_by_name():
    """projectScene/ holds ten scene files: eight parse (triangle meshes, the sphere of scene.pbrt, three volume kinds); the two
    rainbow*_png files put an image-map Texture on a wall (textures are out of scope) and are refused by that name."""
    import glob
    ok, refused = [], []
    for f in sorted(glob.glob(os.path.join(REF_SCENES, "*.pbrt"))):
        try:
            d = ps.load(f)
            abi.SceneHolder(d)
            ok.append(os.path.basename(f))
        except ps.Unsupported as e:
            assert "directive Texture" in str(e), (f, str(e))
            refused.append(os.path.basename(f))
    assert len(ok) == 8 and sorted(refused) == ["rainbow2_png.pbrt", "rainbow_png.pbrt"], (ok, refused)


def test_transform_directives(tmp_path):
    """LookAt / Transform / ConcatTransform / TransformBegin / ReverseOrientation / Scale with a mirror."""
    f = tmp_path / "t.pbrt"
    f.write_text('''
LookAt 1 2 3  1 2 4  0 1 0
Camera "perspective" "float fov" [45]
WorldBegin
TransformBegin
  Transform [1 0 0 0  0 1 0 0  0 0 1 0  2 3 4 1]
  ConcatTransform [2 0 0 0  0 2 0 0  0 0 2 0  0 0 0 1]
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0  1 0 0  0 1 0]
TransformEnd
AttributeBegin
  Scale -1 1 1
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0  1 0 0  0 1 0]
  ReverseOrientation
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0  1 0 0  0 1 0]
AttributeEnd
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0  1 0 0  0 1 0]
WorldEnd
''')
    d = ps.load(str(f))
    P = d["tris.p"].reshape(-1, 3, 3)
    np.testing.assert_array_equal(P[0], [[2, 3, 4], [4, 3, 4], [2, 5, 4]])          # translate (file holds the transpose), then scale 2
    np.testing.assert_array_equal(P[1], [[0, 0, 0], [-1, 0, 0], [0, 1, 0]])         # mirrored
    np.testing.assert_array_equal(P[3], [[0, 0, 0], [1, 0, 0], [0, 1, 0]])          # AttributeEnd restored the CTM
    assert list(d["tris.flip"]) == [0, 1, 0, 0]                                      # handedness swap; ReverseOrientation undoes it; restored
    c2w = d["camera.c2w"].reshape(4, 4)
    np.testing.assert_allclose(c2w[:3, 3], [1, 2, 3], atol=1e-6)                     # camera at the LookAt position, looking down +z
    np.testing.assert_allclose(c2w[:3, 2], [0, 0, 1], atol=1e-6)
    assert float(d["camera.fov"][0]) == 45.0
    assert list(d["mats.kind"]) == [0] and np.all(d["mats.kd"] == np.float32(0.5))   # the default matte material, Kd = Spectrum(0.5)


def test_what_is_not_covered_is_refused_by_name(tmp_path):
    for text, what in [('Camera "perspective"\nWorldBegin\nShape "cylinder" "float radius" [1]\nWorldEnd', 'Shape "cylinder"'),
                       ('Camera "orthographic"', 'Camera "orthographic"'),
                       ('Camera "perspective"\nWorldBegin\nTexture "t" "color" "constant"\nWorldEnd', "directive Texture"),
                       ('Camera "perspective"\nWorldBegin\nLightSource "infinite"\nWorldEnd', 'LightSource "infinite"')]:
        f = tmp_path / "u.pbrt"
        f.write_text(text)
        with pytest.raises(ps.Unsupported) as e:
            ps.load(str(f))
        assert what in str(e.value)


def test_omitted_parameters_take_the_reference_defaults(tmp_path):
    """Every integrator parameter left out: CreatePhotonVolumeIntegrator (integrators/photonvolume.cpp:224-229: stepsize 1,
    nused 250, maxdist 0.1), CreatePhotonShooter (core/photonshooter.cpp:529-548: volumephotons 0, stepsize 0.1,
    maxphotondepth 5, causticphotons 20000, indirectphotons 10000, finalgather true), CreatePhotonMapSurfaceIntegrator
    (integrators/photonmap.cpp:322-333: nused 50, maxspeculardepth 5, finalgather true, finalgathersamples 32, maxdist 0.1,
    gatherangle 10) -- and the same values as pvol_default_params hands the C ABI."""
    f = tmp_path / "d.pbrt"
    f.write_text('Camera "perspective"\nSurfaceIntegrator "photonmap"\nVolumeIntegrator "photonvolume"\nWorldBegin\n'
                 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0  1 0 0  0 1 0]\nWorldEnd\n')
    d = ps.load(str(f))
    np.testing.assert_array_equal(d["params.f"], np.array([1.0, 0.1, 0.1], np.float32))
    assert list(d["params.i"]) == [250, 0, 5, 20000, 10000, 1]
    assert list(d["surf.params.i"]) == [50, 5, 1, 32]
    np.testing.assert_array_equal(d["surf.params.f"], np.array([0.1, 10.0], np.float32))
    p = abi.params_from_blob(d)
    assert (p.n_used, p.n_volume_photons, p.max_photon_depth, p.n_caustic_photons, p.n_indirect_photons, p.final_gather) == (250, 0, 5, 20000, 10000, 1)
    assert abs(p.step_size - 1.0) < 1e-9 and abs(p.max_dist - 0.1) < 1e-7 and abs(p.shooter_step_size - 0.1) < 1e-7


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="the reference tree is not on this machine")
def test_reference_scene_parameters_as_written():
    """pinkfloyd.pbrt states the volume integrator's nused (500) and leaves the shooter's stepsize out (0.1)."""
    d = ps.load(os.path.join(REF_SCENES, "pinkfloyd.pbrt"))
    assert int(d["params.i"][0]) == 500 and int(d["surf.params.i"][0]) == 50
    assert abs(float(d["params.f"][2]) - 0.1) < 1e-7
