"""Debug aid: render a golden case with PVOL_TILE_WAVES = 1 and W and show where the two part ways."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_render as T
name = sys.argv[1] if len(sys.argv) > 1 else "vh"
W = sys.argv[2] if len(sys.argv) > 2 else "4"
res = {}
for w in ("1", W):
    os.environ["PVOL_TILE_WAVES"] = w
    pv, s, p, cam, film, smp, c, old = T._make(name)
    n = len(c["samples.time"])
    r = T._render(torch, pv, cam, film, smp, c["tasks"], n)
    pv.close()
    res[w] = r
    print("waves", w, "end_draw", r["streams"]["end_draw"], "ref", c["task.end_draw"])
    bad = np.nonzero(r["xy"].ravel() != c["samples.image"])[0]
    print("   first xy mismatch", bad[:1], "of", r["xy"].size, " rng_skip[:12]", r["rays"]["rng_skip"][:12], "ref", c["rays.skip"][:12])
a, b = res["1"], res[W]
print("maxt equal", np.array_equal(a["rays"]["maxt"], b["rays"]["maxt"]))
# synthetic: 256 spp (G = 4 sample groups) -- device vs device
for spp in (256, 64, 4):
    out = {}
    for w in ("1", "101", "2", "4", "8", "16"):
        os.environ["PVOL_TILE_WAVES"] = w
        pv, s, p, cam, film, smp, c, old = T._make("vh")
        cam2 = T.abi.perspective_camera(70.0, 16, 8, s["camera.c2w"])
        film2 = T.abi.make_film(16, 8, T._pvol().gaussian_filter_table())
        smp2 = T.abi.make_sampler(16, 8, spp, 8)
        tasks = np.arange(8, dtype=np.uint32)
        n = int(T._pvol().render_sample_count(smp2, tasks))
        r = T._render(torch, pv, cam2, film2, smp2, tasks, n)
        pv.close()
        out[w] = r
        d = np.nonzero((r["xy"] != out["1"]["xy"]).any(axis=1))[0]
        fl = np.nonzero(r["rays"]["flags"] != out["1"]["rays"]["flags"])[0]
        mx = np.nonzero(r["rays"]["maxt"] != out["1"]["rays"]["maxt"])[0]
        print("   first xy mismatch sample", d[:1], "first flags mismatch sample", fl[:1], "first maxt mismatch", mx[:1])
        print("spp", spp, "waves", w, "end_draw", r["streams"]["end_draw"][:4], "xy equal to w=1:", np.array_equal(r["xy"], out["1"]["xy"]),
              "skip equal:", np.array_equal(r["rays"]["rng_skip"], out["1"]["rays"]["rng_skip"]))
