"""Live rays, node fetches and primitive tests per bounce of one scene (instrumented kernels, depth d minus depth d-1),
and what the scattered rays of the first bounces hit."""
import sys
sys.path.insert(0, 'raytracer-rs_amd')
import numpy as np, torch, rtamd
name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
aspect = 1.0 if name.startswith(("cornell", "final")) else 16 / 9
w = 800 if aspect == 1.0 else 1920
h = rtamd.image_height(w, aspect)
ctx = rtamd.Context(0)
info = ctx.load_scene(name, seed=1)
print(info)
cam = ctx.preset_camera(aspect)
ctx.set_pipeline(1)
ctx.profile(False, True)
prev = (0, 0, 0)
for d in list(range(1, 13)) + [16, 24, 32, 50]:
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    _, st = ctx.render(w, h, cam, spp, depth=d, accum=acc)
    cur = (st["rays"], st["node_visits"], st["prim_tests"])
    dr = cur[0] - prev[0]
    print(f"depth {d:2d}: rays so far {cur[0]/1e6:8.2f}M  new {dr/1e6:7.2f}M  nodes/ray {(cur[1]-prev[1])/max(dr,1):6.2f}  prims/ray {(cur[2]-prev[2])/max(dr,1):5.2f}", flush=True)
    prev = cur
