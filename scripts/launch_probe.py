"""Per-launch kernel times of one frame (RTAMD_DEBUG_LIVE=1 prints them): scripts/launch_probe.py scene spp"""
import os, sys
os.environ["RTAMD_DEBUG_LIVE"] = "1"
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
name, spp = sys.argv[1], int(sys.argv[2])
aspect = 1.0 if name.startswith(("cornell", "final")) else 16 / 9
w = 800 if aspect == 1.0 else 1920
h = rtamd.image_height(w, aspect)
ctx = rtamd.Context(0)
ctx.load_scene(name, seed=1)
cam = ctx.preset_camera(aspect)
acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
ctx.render(w, h, cam, 4, accum=acc)
ctx.profile(events=True)
acc.zero_()
print("==== timed frame", file=sys.stderr, flush=True)
_, st = ctx.render(w, h, cam, spp, accum=acc)
print(st)
