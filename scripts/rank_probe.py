"""Renders one rank's share of a frame a few times (for kernel traces of the per-rank launch sequence).
   python scripts/rank_probe.py <scene> <width> <aspect> <spp> <rank> <nranks> [reps]"""
import sys, time
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
name, w, aspect, spp, rank, nranks = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
ctx = rtamd.Context(0)
ctx.load_scene(name, seed=1)
h = rtamd.image_height(w, aspect)
cam = ctx.preset_camera(aspect)
acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
ctx.render(w, h, cam, spp, accum=acc, rank=rank, nranks=nranks)
torch.cuda.synchronize()
ctx.profile(events=True)
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter()
    _, st = ctx.render(w, h, cam, spp, accum=acc, rank=rank, nranks=nranks)
    best = min(best, time.perf_counter() - t0)
dt = best
if True:
    print(f"{name} rank {rank}/{nranks}: {dt*1e3:.2f} ms, {st['rays']/dt/1e6:.0f} Mrays/s, launches {st['bounce_launches']} | trace {st['trace_ms']:.2f} shade {st['shade_ms']:.2f} "
          f"settled {st.get('settled_ms', 0):.2f} bounce {st['bounce_ms']:.2f} tail {st['tail_ms']:.2f} accum {st['accumulate_ms']:.2f}", flush=True)
