"""World-level BVH build: host (binned SAH) against device (linear BVH) -- build time and trace rate.
   python scripts/build_probe.py <n_spheres> <width> <spp>"""
import sys, time
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
n = int(sys.argv[1]); w = int(sys.argv[2]); spp = int(sys.argv[3])
aspect = 16 / 9
h = rtamd.image_height(w, aspect)
for builder in (1, 0):
    ctx = rtamd.Context(0)
    ctx.set_builder(builder)
    t0 = time.perf_counter()
    info = ctx.load_scene("synth_spheres", seed=1, n_synth=n)
    dt = time.perf_counter() - t0
    print("builder %d: load+flatten+commit %.2f s, commit %.3f s, world tree %.3f s | nodes %d depth %d stack %d" % (
        builder, dt, info["build_seconds"], info["tree_seconds"], info["n_nodes"], info["max_depth"], info["stack_entries"]), flush=True)
    cam = ctx.preset_camera(aspect)
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    ctx.render(w, h, cam, 1, accum=acc)
    for pipe in (0, 2):
        ctx.set_pipeline(pipe)
        ctx.profile(True, False)
        acc.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, st = ctx.render(w, h, cam, spp, accum=acc)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("   pipeline %d: %.1f ms, %.1f Mrays/s, trace %.1f shade %.1f bounce %.1f ms" % (pipe, dt * 1e3, st["rays"] / dt / 1e6, st["trace_ms"], st["shade_ms"], st["bounce_ms"]), flush=True)
    ctx.profile(False, True)
    _, st = ctx.render(w, h, cam, 1, accum=acc)
    print("   nodes/ray %.2f prims/ray %.2f" % (st["node_visits"] / st["rays"], st["prim_tests"] / st["rays"]), flush=True)
    ctx.close()
