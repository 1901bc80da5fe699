"""Does the order of the rays matter for a tree far beyond the caches?  Bounce rays of the 10 M-sphere scene
(made on the host from the hits of the camera rays) are intersected in pixel order, sorted by the Morton code
of their origin, and shuffled.   python scripts/c5_sort_probe.py [n_spheres] [width]"""
import sys, time
sys.path.insert(0, 'raytracer-rs_amd')
import numpy as np, torch, rtamd

n_s = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
aspect = 16 / 9
h = rtamd.image_height(w, aspect)
ctx = rtamd.Context(0)
ctx.load_scene("synth_spheres", seed=1, n_synth=n_s)
cam = ctx.preset_camera(aspect)
rng = np.random.default_rng(7)


def unit_vectors(n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def morton(o):
    lo, hi = o.min(axis=0), o.max(axis=0)
    q = ((o - lo) / (hi - lo).max() * 1023.0).astype(np.uint32)
    def spread(x):
        x = x & 0x3FF
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def timed(rays, label):
    dev = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    ctx.intersect(dev, outputs=("prim", "t"))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = ctx.intersect(dev, outputs=("prim", "t"))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"   {label:28s} {best*1e3:8.2f} ms  {rays.shape[1]/best/1e6:8.1f} Mrays/s", flush=True)
    return out


jj, ii = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
u = ((ii + 0.5) / (w - 1)).ravel()
v = ((h - 1 - jj + 0.5) / (h - 1)).ravel()
org = np.array(cam.origin, dtype=np.float32)
d = (np.array(cam.lower_left_corner, dtype=np.float32)[None, :] + u[:, None] * np.array(cam.horizontal, dtype=np.float32)[None, :]
     + v[:, None] * np.array(cam.vertical, dtype=np.float32)[None, :] - org[None, :])
rays = np.empty((7, u.size), dtype=np.float32)
rays[0:3] = org[:, None]
rays[3:6] = d.T
rays[6] = 0.0
for bounce in range(4):
    n = rays.shape[1]
    print(f"bounce {bounce}: {n} rays", flush=True)
    timed(rays, "pixel order")
    if bounce:
        order = np.argsort(morton(rays[0:3].T), kind="stable")
        timed(rays[:, order], "sorted by origin (10 bits/axis)")
        timed(rays[:, rng.permutation(n)], "shuffled")
    full = ctx.intersect(torch.from_numpy(rays).cuda(), outputs=("prim", "t", "normal", "point"))
    hit = (full["prim"].cpu().numpy() >= 0)
    p = full["point"].cpu().numpy()[hit]
    nrm = full["normal"].cpu().numpy()[hit]
    if p.shape[0] < 1000:
        break
    rays = np.empty((7, p.shape[0]), dtype=np.float32)
    rays[0:3] = p.T
    rays[3:6] = (nrm + unit_vectors(p.shape[0])).T
    rays[6] = 0.0
ctx.close()
