"""One line per scene: per-kernel HIP-event times of a frame at the default pipeline (timing experiments)."""
import sys, time, os
sys.path.insert(0, 'raytracer-rs_amd')
import torch, rtamd
tag = os.environ.get("RTAMD_LIB_DIR", "lib").split("/")[-1]
for arg in sys.argv[1:]:
    name, spp = (arg.split(":") + ["64"])[:2]
    spp = int(spp)
    aspect = 1.0 if name.startswith(("cornell", "final")) else 16 / 9
    w = 800 if aspect == 1.0 else 1920
    h = rtamd.image_height(w, aspect)
    ctx = rtamd.Context(0)
    ctx.load_scene(name, seed=1)
    cam = ctx.preset_camera(aspect)
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    ctx.render(w, h, cam, 4, accum=acc)
    ctx.profile(events=True)
    best = None
    for _ in range(3):
        acc.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, st = ctx.render(w, h, cam, spp, accum=acc)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, st)
    dt, st = best
    print(f"{tag:14s} {name:18s} {spp:4d} spp {dt*1e3:8.2f} ms {st['rays']/dt/1e6:8.1f} Mrays/s | trace {st['trace_ms']:7.2f} shade {st['shade_ms']:7.2f} "
          f"settled {st.get('settled_ms', 0.0):7.2f} bounce {st['bounce_ms']:7.2f} tail {st['tail_ms']:5.2f} paths {st['paths_ms']:7.2f} | rays {st['rays']} settled {st['settled_rays']} | sum {acc.sum().item():.6e}", flush=True)
    if os.environ.get("VISITS") == "1":
        ctx.profile(events=False, count_visits=True)
        _, sv = ctx.render(w, h, cam, int(os.environ.get("VISITS_SPP", "4")), accum=acc)
        print(f"               nodes/ray {sv['node_visits']/max(sv['rays'],1):.3f} prims/ray {sv['prim_tests']/max(sv['rays'],1):.3f} stack {ctx.info()['stack_entries']}", flush=True)
    ctx.close()
