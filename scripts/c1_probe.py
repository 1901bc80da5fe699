import sys, time, os
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
ctx = rtamd.Context(0)
ctx.load_scene("random_world", seed=1)
w, h = 400, 225
cam = ctx.preset_camera(16/9)
acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
ctx.render(w, h, cam, 10, accum=acc)
ctx.profile(events=True)
best = 1e9
for _ in range(20):
    acc.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter(); _, st = ctx.render(w, h, cam, 10, accum=acc); dt = time.perf_counter() - t0
    best = min(best, dt)
print(f"C1 {best*1e3:.3f} ms | trace {st['trace_ms']:.3f} shade {st['shade_ms']:.3f} bounce {st['bounce_ms']:.3f} tail {st['tail_ms']:.3f} raygen {st['raygen_ms']:.3f} accum {st['accumulate_ms']:.3f} launches {st['bounce_launches']} tail_rays {st['tail_rays']} rays {st['rays']}")
ctx.profile(events=False)
best = 1e9
for _ in range(50):
    acc.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter(); _, st = ctx.render(w, h, cam, 10, accum=acc); dt = time.perf_counter() - t0
    best = min(best, dt)
print(f"C1 without event timing {best*1e3:.3f} ms")
