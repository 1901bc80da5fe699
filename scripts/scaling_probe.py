#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU: renders each rank's share of the frame (tile % N == rank) one rank
after the other and reports max-over-ranks time per N, i.e. what N GPUs would take without the exchange.
   python scripts/scaling_probe.py [workload] [steps]
SPP=n overrides the workload's samples per pixel.
FRAMES_AT_ONCE=1 renders a rank's shares of the step's frames concurrently (one stream and host thread each), as
bench.py does when the step is split over several ranks.
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "raytracer-rs_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c2_textures"
at_once = os.environ.get("FRAMES_AT_ONCE", "0") == "1"  # as bench.py does for N > 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames_cfg, width, aspect, spp = bench.WORKLOADS[workload]
spp = int(os.environ.get("SPP", spp))  # (a smaller sample count makes every rank's launches small)
t1 = None
for n in (1, 2, 4, 8):
    worst = 0.0
    rays = 0
    for r in range(n):
        frames = [bench.Frame(name, width, aspect, spp, r, n, 0, 10_000_000, own_stream=at_once and n > 1) for name in frames_cfg]
        for f in frames:
            f.gather = None
            f.render()
        torch.cuda.synchronize()
        pool = ThreadPoolExecutor(max_workers=len(frames)) if at_once and n > 1 else None
        t0 = time.perf_counter()
        for _ in range(steps):
            if pool is None:
                for f in frames:
                    st = f.render()
                    rays += st["rays"]
            else:
                for f in frames:
                    f.clear()
                torch.cuda.current_stream().synchronize()
                for st in pool.map(lambda f: f.trace(), frames):
                    rays += st["rays"]
        torch.cuda.synchronize()
        worst = max(worst, (time.perf_counter() - t0) / steps)
        for f in frames:
            f.ctx.close() if hasattr(f.ctx, "close") else None
        del frames
        if n == 8 and r >= 2:  # three ranks of eight are enough to see the spread
            break
    if n == 1:
        t1 = worst
    print(f"{workload}: N={n} max-rank step {worst * 1e3:8.2f} ms  efficiency {t1 / (n * worst):.3f}", flush=True)
