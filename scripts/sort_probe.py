"""How much does the order of incoherent rays matter to the tree walk?  Builds the first scattered rays of a frame on
the device (camera ray -> hit -> normal + random unit vector), then times rt_intersect_batch (one ray per lane) on
them in pixel order, shuffled, and ordered by direction octant inside tiles of 256 / 1024 / 4096 / all rays."""
import sys, time
sys.path.insert(0, 'raytracer-rs_amd')
import numpy as np, torch, rtamd
name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
aspect = 1.0 if name.startswith(("cornell", "final")) else 16 / 9
w = 1600 if aspect == 1.0 else 1920
h = rtamd.image_height(w, aspect)
ctx = rtamd.Context(0)
ctx.load_scene(name, seed=1)
ctx.use_torch_stream()
cam = ctx.preset_camera(aspect)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
jj, ii = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
u = ((ii + torch.rand((h, w), device=dev, generator=g)) / (w - 1)).reshape(-1)
v = (((h - 1 - jj) + torch.rand((h, w), device=dev, generator=g)) / (h - 1)).reshape(-1)
f3 = lambda a: torch.tensor(list(a), device=dev, dtype=torch.float32)
o = f3(cam.origin)[:, None].expand(3, u.numel())
d = f3(cam.lower_left_corner)[:, None] + u * f3(cam.horizontal)[:, None] + v * f3(cam.vertical)[:, None] - f3(cam.origin)[:, None]
rays = torch.cat([o, d, torch.rand((1, u.numel()), device=dev, generator=g)]).contiguous()
hit = ctx.intersect(rays, seed=1)
ok = hit["prim"] >= 0
n = hit["normal"][ok].T
rnd = torch.randn((3, n.shape[1]), device=dev, generator=g)
rnd = rnd / rnd.norm(dim=0, keepdim=True)
sec = torch.cat([hit["point"][ok].T, n + rnd, rays[6:7, ok]]).contiguous()
N = sec.shape[1]
print(name, "secondary rays", N)

def run(r, label):
    r = r.contiguous()
    outs = ("prim", "t")
    ctx.intersect(r, seed=1, outputs=outs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.intersect(r, seed=1, outputs=outs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{label:32s} {dt*1e3:8.3f} ms  {N/dt/1e6:9.1f} Mrays/s", flush=True)

octant = ((sec[3] < 0).long() | ((sec[4] < 0).long() << 1) | ((sec[5] < 0).long() << 2))
run(sec, "pixel order")
run(sec[:, torch.randperm(N, device=dev, generator=g)], "shuffled")
for tile in (256, 1024, 4096, 65536, 1 << 30):
    key = (torch.arange(N, device=dev) // tile) * 8 + octant
    run(sec[:, torch.argsort(key, stable=True)], f"octant inside tiles of {tile}")
# finer direction bins: octant + dominant axis (24 bins)
dom = sec[3:6].abs().argmax(dim=0)
for tile in (1024, 65536):
    key = (torch.arange(N, device=dev) // tile) * 24 + octant * 3 + dom
    run(sec[:, torch.argsort(key, stable=True)], f"octant+axis inside tiles of {tile}")
