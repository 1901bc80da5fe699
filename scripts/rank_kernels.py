"""Per-kernel times of one rank's share of a frame against the whole frame (where does the strong-scaling loss come from?).
   python scripts/rank_kernels.py <scene> <width> <aspect> <spp>"""
import sys, time
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
name, w, aspect, spp = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
tile = int(sys.argv[5]) if len(sys.argv) > 5 else 32
ctx = rtamd.Context(0)
ctx.load_scene(name, seed=1)
h = rtamd.image_height(w, aspect)
cam = ctx.preset_camera(aspect)
acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
ctx.profile(events=True)
import os
cases = ((0, 1, spp),) if os.environ.get('ONLY_WHOLE') else ((0, 1, spp), (0, 2, spp), (1, 2, spp), (0, 8, spp), (3, 8, spp), (7, 8, spp))
for rank, nranks, s in cases:
    ctx.render(w, h, cam, s, accum=acc, rank=rank, nranks=nranks, tile_size=tile)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, st = ctx.render(w, h, cam, s, accum=acc, rank=rank, nranks=nranks, tile_size=tile)
    dt = time.perf_counter() - t0
    r = st["rays"]
    print(f"tile {tile:2d} {name} rank {rank}/{nranks} {s:5d} spp: {dt*1e3:8.2f} ms {r/dt/1e6:8.0f} Mrays/s | per Grays: trace {st['trace_ms']/r*1e9:6.2f} shade {st['shade_ms']/r*1e9:6.2f} "
          f"bounce {st['bounce_ms']/r*1e9:6.2f} tail {st['tail_ms']/r*1e9:5.2f} accum {st['accumulate_ms']/r*1e9:5.2f} ms | batches {st['batches']} launches {st['bounce_launches']}", flush=True)
