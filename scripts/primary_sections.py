"""Wave cycles per section of the fused kernel on camera rays only (instrumented build, depth 1)."""
import os, sys
os.environ["RTAMD_SECTION_CYCLES"] = "1"
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
for name in sys.argv[1:] or ["random_world", "final_scene"]:
    aspect = 1.0 if name.startswith(("cornell", "final")) else 16 / 9
    w = 800 if aspect == 1.0 else 1920
    ctx = rtamd.Context(0)
    ctx.load_scene(name, seed=1)
    ctx.set_pipeline(1)
    ctx.profile(True, True)
    h = rtamd.image_height(w, aspect)
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    sys.stderr.write(name + ": ")
    sys.stderr.flush()
    _, st = ctx.render(w, h, ctx.preset_camera(aspect), 16, depth=1, accum=acc)
    print(name, st["rays"], "nodes/ray %.2f prims/ray %.2f bounce_ms %.2f" % (st["node_visits"] / st["rays"], st["prim_tests"] / st["rays"], st["bounce_ms"]), flush=True)
