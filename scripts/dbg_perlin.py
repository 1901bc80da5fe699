import os, sys
sys.path.insert(0, 'raytracer-rs_amd'); sys.path.insert(0, 'tests')
import numpy as np, rtamd
w, h, spp = 160, 90, 2
out = {}
for mode in ("1", "0"):
    os.environ["RTAMD_LDS_TABLES"] = mode
    for tail in (0, 1 << 30):
        ctx = rtamd.Context(0)
        ctx.set_tail_paths(tail)
        ctx.load_scene("two_perlin_spheres", seed=1)
        img, st = ctx.render(w, h, ctx.preset_camera(16 / 9), spp, seed=1)
        out[(mode, tail)] = img
        ctx.close()
ref = out[("0", 0)]
for k, v in out.items():
    d = np.abs(v - ref).max(axis=2)
    print(k, "max diff", d.max(), "pixels differing", (d > 1e-6).sum(), "of", d.size, "rows", np.unique(np.nonzero(d > 1e-6)[0])[:10])
