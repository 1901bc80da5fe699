"""Small fixed workload for counter collection: one scene, one pipeline, a few frames."""
import sys
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
name = sys.argv[1]; pipe = int(sys.argv[2]); w = int(sys.argv[3]); spp = int(sys.argv[4])
aspect = 1.0 if name.startswith(("cornell", "final")) else 16 / 9
ctx = rtamd.Context(0)
n_synth = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx.load_scene(name, seed=1, **({'n_synth': n_synth} if n_synth else {}))
ctx.set_pipeline(pipe)
h = rtamd.image_height(w, aspect)
acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
_, st = ctx.render(w, h, ctx.preset_camera(aspect), spp, accum=acc)
print(name, pipe, st["rays"], st["render_ms"])
