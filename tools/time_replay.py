#!/usr/bin/env python3
"""Times the resolve/replay path on a synthetic uniform photon map: pinkfloyd (C3-like: k = 500, two lights) by default,
or the scene named on the command line (volumescene_grid16: C4-like VolumeGrid).
    PVOL_LIB=cs348b-pbrt_amd/libpvol_w2.so python tools/time_c3.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import measure_configs as mc
abi, blob, pvol = mc.abi, mc.blob, mc.pvol
scene_name = sys.argv[1] if len(sys.argv) > 1 else "pinkfloyd"
scene = blob.load(os.path.join(mc.GOLD, "scene_%s.bin" % scene_name))
params = abi.params_from_blob(scene)
rng = np.random.default_rng(1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000000
ext = scene["vol.extent"].astype(np.float64)
v2w = scene["vol.v2w"].reshape(4, 4).astype(np.float64)
pv_ = ext[:3] + (ext[3:] - ext[:3]) * rng.random((n, 3))
P = (pv_ @ v2w[:3, :3].T + v2w[:3, 3]).astype(np.float32)
W = rng.normal(size=(n, 3)).astype(np.float32); W /= np.linalg.norm(W, axis=1, keepdims=True)
A = (rng.random((n, 30)) * 1e-4).astype(np.float32)
pv = pvol.PhotonVolume(params)
pv.set_scene(abi.SceneHolder(scene))
pv.upload_photons(P, W, A)
rays, streams = mc.rays_for(scene, 240, 135, 2, 7)
pv.li(rays[:2048], abi.make_streams(np.array([0], np.uint32), np.array([2048], np.uint32)), abi.OUT_XYZ)
pv.kernel_time_ms(reset=True)
out, draws = pv.li(rays, streams, abi.OUT_XYZ)
kms, _ = pv.kernel_time_ms()
print(json.dumps({"scene": scene_name, "lib": os.environ.get("PVOL_LIB", "default"), "kernel": pv.march_kernel_name(), "rays": len(rays), "kernel_ms": kms,
                  "Msamples_per_s": len(rays) / kms / 1e3, "mean_draws": float(draws.mean()), "nused": params.n_used, "checksum": float(out[:, :3].sum())}))
pv.close()
