"""Lane use of the persistent trace kernel (instrumented build): split pipeline, prints the [rtamd] k_trace lanes line."""
import os, sys
os.environ["RTAMD_TRACE_LANES"] = "1"
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
for name in sys.argv[1:] or ["final_scene", "random_world"]:
    aspect = 1.0 if name.startswith(("cornell", "final")) else 16 / 9
    w = 800 if aspect == 1.0 else 1920
    h = rtamd.image_height(w, aspect)
    ctx = rtamd.Context(0)
    ctx.load_scene(name, seed=1)
    ctx.set_pipeline(0)
    ctx.profile(False, True)
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    sys.stderr.write(name + ": ")
    sys.stderr.flush()
    _, st = ctx.render(w, h, ctx.preset_camera(aspect), 16, accum=acc)
    print(name, "rays", st["rays"], "nodes/ray %.2f prims/ray %.2f" % (st["node_visits"] / st["rays"], st["prim_tests"] / st["rays"]), flush=True)
    ctx.close()
