"""The reference-side binding's Scene -> pvol_scene walk (integration/hip_flatten.h), run on the reference's OWN objects.

`oracle/_ref/ref_capture shimscene NAME` builds a BASELINE scene through the reference's Create*() functions and its
CreateBVHAccelerator, hands the live `Scene *` -- as SamplerRenderer would hand it to VolumeIntegrator::Preprocess -- to
HipFlattenScene, and writes what came out.  tests/golden/scene_*.bin were written from the builders' own shape lists *before*
the accelerator reordered anything (ref_capture `scene`); the two must agree array for array.

Order and equal-t ties: BVHAccel keeps its primitives in leaf order (accelerators/bvh.cpp:226-237), which is not the order
of the scene file; the walk restores creation order through Shape::shapeId, so the flattened triangle array is *identical*
to the pre-accelerator one -- and with it the device's tie rule (same t: the triangle latest in scene order wins, as a
linear scan's `t > maxt` rejection has it).  Materials are numbered by first use in that order, so indices may be a
permutation of the builders'; they are compared through their content.

CPU only, and only where oracle/_ref exists (the container that holds /root/reference)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, blob, load_scene

TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_capture")
pytestmark = pytest.mark.skipif(not os.path.exists(TOOL), reason="oracle/_ref/ref_capture is not built on this machine")

SCENES = ["volumescene_h", "volumescene_rainbow", "volumescene_grid16", "pinkfloyd", "shootbench", "meshroom", "spherescene", "sphereroom"]
MAT_KEYS = ("mats.kind", "mats.kd", "mats.kr", "mats.kt", "mats.ior", "mats.vn")
NOT_SCENE = ("params.f", "params.i", "camera.c2w", "camera.fov", "film")   # integrator / camera parameters: not the flattening's business


def _shim(name, tmp_path):
    out = str(tmp_path / ("shim_%s.bin" % name))
    subprocess.run([TOOL, "shimscene", name, out], check=True, timeout=120)
    return blob.load(out)


def _materials(d):
    n = len(d["mats.kind"])
    rows = [np.concatenate([np.asarray(d[k]).reshape(n, -1).astype(np.float32).view(np.uint32) for k in MAT_KEYS], axis=1)][0]
    return rows


@pytest.mark.parametrize("name", SCENES)
def test_live_scene_flattens_to_the_golden_scene(name, tmp_path):
    got, ref = _shim(name, tmp_path), load_scene(name)
    for k, v in ref.items():
        if k in NOT_SCENE or k in MAT_KEYS or k.endswith(".material"):
            continue
        assert k in got, k
        a, b = np.asarray(got[k]), np.asarray(v)
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert a.tobytes() == b.tobytes(), k      # bit for bit, triangles IN ORDER
    # materials through the index: same content per triangle / sphere, same set
    gm, rm = _materials(got), _materials(ref)
    assert sorted(map(bytes, gm)) == sorted(map(bytes, rm))
    for key in ("tris.material", "spheres.material"):
        if key in ref:
            np.testing.assert_array_equal(gm[np.asarray(got[key])], rm[np.asarray(ref[key])], err_msg=key)


def test_the_accelerator_reorders_and_the_walk_undoes_it(tmp_path):
    """Why the walk sorts: the BVH's leaf order is NOT creation order on the 966-triangle scene (nor on the 10-triangle one)."""
    for name in ("meshroom", "pinkfloyd"):
        ids = np.asarray(_shim(name, tmp_path)["bvh.leaf_shape_ids"])
        assert len(ids) == len(load_scene(name)["tris.material"])
        assert len(np.unique(ids)) == len(ids)
        assert not (np.diff(ids) > 0).all()
