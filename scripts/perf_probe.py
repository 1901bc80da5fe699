"""Per-scene timing breakdown (HIP events per kernel) at reduced sample counts, both pipelines."""
import sys, time
sys.path.insert(0, 'raytracer-rs_amd')
import numpy as np, torch, rtamd
cases = [("earth", 1920, 16/9, 64), ("two_spheres", 1920, 16/9, 64), ("two_perlin_spheres", 1920, 16/9, 64),
         ("random_world", 1920, 16/9, 64), ("cornell_box", 1024, 1.0, 64), ("cornell_smoke", 1024, 1.0, 64),
         ("final_scene", 800, 1.0, 64)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, w, aspect, spp in cases:
    ctx = rtamd.Context(0)
    info = ctx.load_scene(name, seed=1)
    h = rtamd.image_height(w, aspect)
    cam = ctx.preset_camera(aspect)
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    for pipe in (0, 1, 2):
        ctx.set_pipeline(pipe)
        ctx.render(w, h, cam, 4, accum=acc)
        ctx.profile(events=True)
        acc.zero_()
        _, st = ctx.render(w, h, cam, spp, accum=acc)
        ctx.profile(events=False, count_visits=True)
        _, sc = ctx.render(w, h, cam, 1, accum=acc)
        ctx.profile(events=False)
        acc.zero_()
        t0 = time.perf_counter(); _, s2 = ctx.render(w, h, cam, spp, accum=acc); dt = time.perf_counter() - t0
        print(f"{name:20s} {('split', 'fused', 'hybrd')[pipe]} {w}x{h}x{spp} rays {st['rays']/1e6:8.1f}M r/p {st['rays']/st['paths']:.2f} | {dt*1e3:8.2f} ms = {s2['rays']/dt/1e6:8.1f} Mrays/s | "
              f"trace {st['trace_ms']:7.2f} shade {st['shade_ms']:7.2f} bounce {st['bounce_ms']:7.2f} raygen {st['raygen_ms']:6.2f} accum {st['accumulate_ms']:6.2f} | batches {st['batches']} launches {st['bounce_launches']} | "
              f"nodes/ray {sc['node_visits']/sc['rays']:.2f} prims/ray {sc['prim_tests']/sc['rays']:.2f} | nodes {info['n_nodes']}", flush=True)
    ctx.close()
