"""Counts rays with non-finite components per sample range (instrumented split pipeline)."""
import os, sys
os.environ["RTAMD_TRACE_LANES"] = "1"
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
name = sys.argv[1]
aspect = 1.0 if name.startswith(("cornell", "final")) else 16 / 9
w = 800 if aspect == 1.0 else 1920
h = rtamd.image_height(w, aspect)
ctx = rtamd.Context(0)
ctx.load_scene(name, seed=1)
ctx.set_pipeline(0)
ctx.profile(False, True)
acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
for s0 in range(0, 256, 32):
    sys.stderr.write(f"samples {s0}..{s0+32}: ")
    sys.stderr.flush()
    _, st = ctx.render(w, h, ctx.preset_camera(aspect), 32, accum=acc, sample0=s0)
    print(s0, st["render_ms"], flush=True)
print("finite image:", bool(torch.isfinite(acc).all()))
