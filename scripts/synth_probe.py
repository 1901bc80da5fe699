"""Synthetic many-sphere scene: build time and Mrays/s of both pipelines."""
import sys, time
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
n = int(sys.argv[1]); w = int(sys.argv[2]); spp = int(sys.argv[3])
aspect = 16 / 9
h = rtamd.image_height(w, aspect)
ctx = rtamd.Context(0)
t0 = time.perf_counter()
info = ctx.load_scene("synth_spheres", seed=1, n_synth=n)
print("load+commit %.2f s" % (time.perf_counter() - t0), {k: info[k] for k in ("n_prims", "n_nodes", "max_depth", "stack_entries", "device_bytes", "build_seconds")}, flush=True)
cam = ctx.preset_camera(aspect)
for pipe in (0, 2):
    ctx.set_pipeline(pipe)
    ctx.profile(True, False)
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    ctx.render(w, h, cam, 1, accum=acc)
    acc.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, st = ctx.render(w, h, cam, spp, accum=acc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pipeline %d: %.1f ms, %.1f Mrays/s, rays/path %.2f, trace %.1f shade %.1f bounce %.1f ms | sum %.9e" % (pipe, dt * 1e3, st["rays"] / dt / 1e6, st["rays"] / st["paths"], st["trace_ms"], st["shade_ms"], st["bounce_ms"], acc.double().sum().item()), flush=True)
ctx.profile(False, True)
_, st = ctx.render(w, h, cam, 1, accum=acc)
print("nodes/ray %.2f prims/ray %.2f" % (st["node_visits"] / st["rays"], st["prim_tests"] / st["rays"]))
