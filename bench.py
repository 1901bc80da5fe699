#!/usr/bin/env python3
"""Throughput of the MI355X render hot path, on the metric BASELINE.json names: Mrays/s.

One "step" renders the named workload once: every frame of the workload at its full size and
sample count, the frame's tiles dealt round-robin over the ranks, each rank's tiles gathered into
the complete frame on every rank with one all_gather (RCCL over xGMI) per frame.  Inputs (scene,
BVH, textures, pixel lists) are resident in HBM before the timed region starts.

The default workload is the scene the reference binary itself renders (main.rs:353-354, 375-380):
the Next Week final scene, 800x800, 10 000 samples per pixel, depth 50 -- the tree-walking path
(BVHNode::intersect) the metric is about.  With one rank the same run also renders every one of
BASELINE.json's five configs once at full size outside the timed steps and lists them under config.others.

  python bench.py                       # 1 GPU, default steps
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is whole-job Mrays/s: rays traced by all ranks (one ray = one
world.intersect call issued by ray_color, primary and scattered) divided by the max-over-ranks
time of the K timed steps.  The same frame is split over the ranks, so scaling is "strong".
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "raytracer-rs_amd"))

import torch  # noqa: E402

import rtamd  # noqa: E402
from rtamd.dist import TileGather  # noqa: E402

# BASELINE.json configs: name -> (frames, width, aspect, spp)
WORKLOADS = {
    # configs[1]: earth sphere with image texture + checker / noise textures, 1920x1080, 256 spp
    "c2_textures": (("earth", "two_spheres", "two_perlin_spheres"), 1920, 16.0 / 9.0, 256),
    # configs[0]: One Weekend random spheres, 400x225, 10 spp (the reference's own CPU-sized case)
    "c1_random_world": (("random_world",), 400, 16.0 / 9.0, 10),
    # configs[2]: random spheres with BVH + moving spheres, 1920x1080, 1024 spp
    "c3_random_world": (("random_world",), 1920, 16.0 / 9.0, 1024),
    # configs[3]: Cornell box + constant-medium volumes, 2048x2048, 4096 spp
    "c4_cornell": (("cornell_box", "cornell_smoke"), 2048, 1.0, 4096),
    # the north star's target scene at the reference binary's own preset (main.rs:375-380): 800x800, 10 000 spp
    "final_scene": (("final_scene",), 800, 1.0, 10000),
    # configs[4]: synthetic 10M-sphere scene, 3840x2160, 1024 spp
    "c5_synth10m": (("synth_spheres",), 3840, 16.0 / 9.0, 1024),
}
MAX_DEPTH = 50
PEAK_HBM_GBPS = 8000.0  # MI355X HBM3E spec peak
COMPUTE_UNITS = 256  # MI355X
DEFAULT_WORKLOAD = "final_scene"
# one frame each, outside the timed steps, when the default workload runs on one rank
OTHER_CONFIGS = ("c1_random_world", "c2_textures", "c3_random_world", "c4_cornell", "c5_synth10m")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Frame:
    """One scene of the workload on this rank: context, camera, accumulation buffer, gather plan."""

    def __init__(self, name, width, aspect, spp, rank, world, device, n_synth, tile=None, gather_on_cpu=False, own_stream=False,
                 exchange="torch"):
        # 16x16 tiles split a frame evenly over up to four ranks (and are the fastest pixel order on one); eight ranks
        # want 8x8 tiles (final_scene 800x800, slowest of eight ranks: 57.4 ms with 16, 54.6 ms with 8, per 1024 spp)
        if tile is None:
            tile = 16 if world <= 4 else 8
        self.name, self.w, self.spp = name, width, spp
        self.h = rtamd.image_height(width, aspect)
        self.rank, self.world = rank, world
        self.ctx = rtamd.Context(device)
        self.info = self.ctx.load_scene(name, seed=1, n_synth=n_synth)
        # own_stream: the context keeps its own HIP stream, so that several frames can render at once (render() then
        # has to be called with the torch stream idle: the accumulation buffer is zeroed on that one)
        self.own_stream = own_stream
        if not own_stream:
            self.ctx.use_torch_stream()
        self.cam = self.ctx.preset_camera(aspect)
        self.accum = torch.zeros((self.h, self.w, 3), dtype=torch.float32, device=f"cuda:{device}")
        self.tile = tile
        # the gloo rehearsal backend has no device collectives: stage the exchange through host memory there
        self.gather_dev = torch.device("cpu") if gather_on_cpu else self.accum.device
        self.gather = None
        self.exchange_kind = "none"
        if world > 1:
            if exchange == "abi" and not gather_on_cpu and self._make_comm(rank, world, device):
                # the exchange behind the C ABI: rt_gather_frame packs, runs ONE ncclAllGather on the context's own RCCL
                # communicator and scatters -- what a host program without torch would call
                self.exchange_kind = "abi"
                self.frame = torch.empty_like(self.accum)
            else:
                self.exchange_kind = "torch"
                self.gather = TileGather(self.w, self.h, rank, world, self.gather_dev, tile)

    def _make_comm(self, rank, world, device):
        """RCCL communicator of this frame's context, made through the C ABI; the 128-byte id travels over
        torch.distributed.  Every step is agreed on by all ranks (a rank that cannot load RCCL must not leave the
        others waiting in ncclCommInitRank): returns False on every rank if any rank failed."""
        dev = f"cuda:{device}"

        def all_ok(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
            return bool(t.item())

        uid, ok = None, True
        try:
            uid = rtamd.comm_unique_id()  # (every rank: loads RCCL; only rank 0's bytes are used)
        except rtamd.RtError as e:
            log(f"rank {rank}: no RCCL for the C-ABI exchange ({e}); falling back to torch.distributed")
            ok = False
        if not all_ok(ok):
            return False
        t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
        torch.distributed.broadcast(t, src=0)
        try:
            self.ctx.comm_create(bytes(t.cpu().tolist()), rank, world)
        except rtamd.RtError as e:
            log(f"rank {rank}: rt_comm_create failed ({e}); falling back to torch.distributed")
            ok = False
        if not all_ok(ok):
            self.ctx.comm_destroy()
            return False
        return True

    def clear(self):
        self.accum.zero_()

    def trace(self, spp=None):
        _, st = self.ctx.render(self.w, self.h, self.cam, spp or self.spp, depth=MAX_DEPTH, seed=1, accum=self.accum,
                                tile_size=self.tile, rank=self.rank, nranks=self.world)
        return st

    def exchange(self):
        # framebuffer exchange: every rank owns disjoint tiles -> one all_gather of the packed tiles
        if self.exchange_kind == "abi":
            if self.own_stream:
                torch.cuda.current_stream().synchronize()
            self.ctx.gather_frame(self.accum, tile_size=self.tile, out=self.frame)
        elif self.gather is not None:
            self.frame = self.gather.gather(self.accum.to(self.gather_dev))

    def render(self, spp=None, exchange=True):
        self.clear()
        if self.own_stream:
            torch.cuda.current_stream().synchronize()
        st = self.trace(spp)
        if exchange:
            self.exchange()
        return st

    def image(self):
        return self.frame if self.exchange_kind != "none" else self.accum



L2_PER_XCD = 4 << 20  # a scene below this stays resident in every XCD's L2 for a whole launch


def scene_bytes_per_ray(info, nodes_per_ray, prims_per_ray):
    """Algorithmic HBM bytes a ray needs from the scene description: 64 B per inner node fetched (two child boxes
    + two references; 128 B where the tree is walked four children at a time, nodes_per_ray then counts those
    fetches) and 32 B per primitive record tested -- unless the whole scene fits the 4 MiB L2 of an XCD.
    Then the launch reads it from HBM once per XCD and every further visit is a cache hit by construction
    (that part is added per launch, see scene_bytes_per_launch), so nothing is charged per ray."""
    if info["device_bytes"] <= L2_PER_XCD:
        return 0.0
    return (128.0 if info.get("bvh_width", 2) == 4 else 64.0) * nodes_per_ray + 32.0 * prims_per_ray


def scene_bytes_per_launch(info):
    return 8.0 * info["device_bytes"] if info["device_bytes"] <= L2_PER_XCD else 0.0


def trace_bytes_per_ray(info, nodes_per_ray, prims_per_ray):
    """Algorithmic bytes one ray moves through k_trace: its ray in (origin+time 16 B, direction+slot 16 B), its
    hit record out (16 B), and the scene bytes of scene_bytes_per_ray."""
    return 32 + 16 + scene_bytes_per_ray(info, nodes_per_ray, prims_per_ray)


def shade_bytes_per_ray(alive_frac):
    """k_shade: path state in (56 B) + hit record (16 B); out: the compacted state of surviving paths (56 B) or
    the radiance of ended ones (16 B).  The hit primitive, material and texture records are cache hits."""
    return 56 + 16 + 56.0 * alive_frac + 16.0 * (1.0 - alive_frac)


def bounce_bytes_per_ray(info, primary_frac, alive_frac, nodes_per_ray, prims_per_ray):
    """k_bounce (fused pipeline): path state in (56 B; a primary ray reads only its 4-B pixel index), the scene
    bytes of scene_bytes_per_ray; out: the compacted state of a surviving path (56 B) or the radiance of an
    ended one (16 B)."""
    return (56.0 * (1.0 - primary_frac) + 4.0 * primary_frac + scene_bytes_per_ray(info, nodes_per_ray, prims_per_ray) +
            56.0 * alive_frac + 16.0 * (1.0 - alive_frac))


def profiled_size_matches(doc, width, spp):
    """A committed profile summary speaks for a run only if it was taken at the same --width / --spp overrides (0 = the
    workload's own size): per-launch figures do not carry over to launches of another size."""
    import re
    cmd = doc.get("command", "")
    def opt(name):
        m = re.search(rf"--{name}[ =](\d+)", cmd)
        return int(m.group(1)) if m else 0
    return opt("width") == (width or 0) and opt("spp") == (spp or 0)


def measured_traffic(workload, kernel, width=0, spp=0):
    """(HBM bytes, file, launches) of a kernel over the profiled run, from the newest committed PMC summary of this workload
    (profiles/*_<workload>_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc
    passes of this script, the read side doubled as gfx950 needs).  None when no summary matches."""
    best = None
    for f in sorted((ROOT / "profiles").glob(f"*_{workload}*_hbm_traffic.json")):
        try:
            doc = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if doc.get("kernel", "").startswith(kernel) and doc.get("bytes_per_step") and profiled_size_matches(doc, width, spp):
            launches = max(doc.get("dispatches_fetch_pass", 0), 1)
            best = (doc["read_bytes"] + doc["write_bytes"], f.name, launches)
    return best


def host_threads():
    """CPU threads this process may use: its affinity mask (the box's CPU share; os.cpu_count() reports the whole
    machine there), capped by OMP_NUM_THREADS when the harness sets it."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS", "")
    if cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


ORACLE_FAST_FLAGS = ("-O3 -fno-semantic-interposition -ftls-model=initial-exec -DRT_ORACLE_FAST -std=c++17 -fPIC -ffp-contract=off "
                     "-fno-fast-math -pthread")


def oracle_for_this_host():
    """(library path, flags) of the oracle build to time.  The committed fast build (oracle/liboracle_fast.so) targets a
    generic AVX2 server because it is compiled in one container and run on another host; when a compiler is at hand
    the same source is rebuilt here with -march=native (the CPU baseline is quoted for a build made for the host it runs on).  The rebuilt library
    must reproduce the checker build's image bit for bit on a small frame, else the committed one is used."""
    sys.path.insert(0, str(ROOT / "tests"))
    import shutil
    import subprocess
    import tempfile
    import orc
    fallback = (orc._FAST_PATH, "-march=x86-64-v3 " + ORACLE_FAST_FLAGS + " (oracle/Makefile, built ahead of the run)")
    cxx = shutil.which("g++")
    if not cxx:
        return fallback
    try:
        out = Path(tempfile.mkdtemp(prefix="rt_oracle_native_")) / "liboracle_native.so"
        cmd = [cxx, "-march=native", *ORACLE_FAST_FLAGS.split(), "-shared", "-o", str(out), str(ROOT / "oracle" / "rt_oracle.cpp")]
        subprocess.run(cmd, check=True, timeout=300, capture_output=True)
        a, b = orc.Scene("final_scene", seed=1), orc.Scene("final_scene", seed=1, lib_path=out)
        img_a, rays_a = a.render(48, 48, 1.0, 2, depth=MAX_DEPTH, seed=1, mode=1)
        img_b, rays_b = b.render(48, 48, 1.0, 2, depth=MAX_DEPTH, seed=1, mode=1)
        a.close()
        b.close()
        if rays_a != rays_b or not np.array_equal(img_a, img_b):
            log("cpu_baseline: the -march=native oracle differs from the checker build; timing the committed fast build instead")
            return fallback
        return out, "-march=native " + ORACLE_FAST_FLAGS + " (rebuilt on this host)"
    except Exception as e:  # no compiler, no temp dir, a time limit: the committed build is timed
        log(f"cpu_baseline: could not rebuild the oracle for this host ({type(e).__name__}); timing the committed fast build")
        return fallback


def cpu_baseline(frames_cfg, width, aspect, seconds_target=20.0):
    """The CPU oracle (a port of the reference; the Rust binary cannot be built here) on a bounded sample of the
    same workload: the same frames at full resolution and one reduced sample count for all of them, the
    reference's own structure (pointer BVH, always-left-then-right walk, one pixel per task), on every host
    thread this process may use.  The program timed is the -O3 build without the checker's visit counters."""
    sys.path.insert(0, str(ROOT / "tests"))
    import orc
    lib_path, flags = oracle_for_this_host()
    threads = host_threads()
    h = rtamd.image_height(width, aspect)
    scenes = [orc.Scene(name, seed=1, lib_path=lib_path) for name in frames_cfg]
    rays_total, sec_total = 0, 0.0
    for sc in scenes:  # one sample per pixel of every frame: sizes the rest
        t0 = time.perf_counter()
        _, rays = sc.render(width, h, aspect, 1, depth=MAX_DEPTH, seed=1, mode=1, threads=threads)
        sec_total += time.perf_counter() - t0
        rays_total += rays
    spp = int(max(1, min(256, seconds_target / max(sec_total, 1e-3))))
    if spp > 1:
        for sc in scenes:
            t0 = time.perf_counter()
            _, rays = sc.render(width, h, aspect, spp - 1, depth=MAX_DEPTH, seed=1, mode=1, s0=1, threads=threads)
            sec_total += time.perf_counter() - t0
            rays_total += rays
    for sc in scenes:
        sc.close()
    return {"value": rays_total / sec_total / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{'+'.join(frames_cfg)} at {width}x{h}, {spp} spp each, depth {MAX_DEPTH}, "
                      f"oracle BVHNode path (oracle/rt_oracle.cpp built {flags}), {sec_total:.1f} s on {threads} threads = "
                      f"this process's CPU affinity mask capped by OMP_NUM_THREADS, not the core count of the machine "
                      f"({os.cpu_count()} logical CPUs)"}


def pmc_profile(workload, kernel, width=0, spp=0):
    """VALU counters of the dominant kernel from the newest committed PMC summary of this workload
    (profiles/*_<workload>_<kernel>_pmc.json: valu_busy = share of SIMD cycles that issue a VALU instruction,
    lane_use = active lanes per VALU instruction / 64).  None when no summary matches."""
    best = None
    for f in sorted((ROOT / "profiles").glob(f"*_{workload}*_{kernel}_pmc.json")):
        try:
            doc = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if "valu_busy" in doc and "lane_use" in doc and profiled_size_matches(doc, width, spp):
            best = {"valu_busy": doc["valu_busy"], "lane_use": doc["lane_use"], "source": f.name}
            # The other unit a tree walk saturates: a vector load whose lanes read scattered places costs the compute
            # unit one clock per active lane (scripts/ta_probe.hip: 0.99 lane-loads per CU clock at best, a quarter of the
            # cost when the four lanes of a quad share an address).  Lanes per load instruction are not counted by any
            # counter collected here; the lanes per VALU instruction stand in for them.
            c = doc.get("counters", {})
            if c.get("SQ_INSTS_VMEM_RD") and doc.get("kernel_ms") and doc.get("clock_GHz"):
                cu_clocks = doc["kernel_ms"] * 1e-3 * doc["clock_GHz"] * 1e9 * COMPUTE_UNITS
                best["vmem_lane_loads_per_cu_clk"] = round(c["SQ_INSTS_VMEM_RD"] * 64.0 * doc["lane_use"] / cu_clocks, 3)
    return best


def kernel_rooflines(frames, stats, steps, workload, use_committed, width=0, spp=0):
    """Per-kernel HBM roofline figures of a set of timed steps.  stats: one list of per-frame rt_render stats per
    step.  Every kernel is charged the bytes it has to move itself (never another formulation's); a scene that
    fits one XCD's L2 is charged once per XCD and launch, a larger one per node fetch and primitive test, and where
    that exceeds what HBM can deliver (upper tree levels served by L2 / Infinity Cache) the committed measured
    traffic takes its place, so that no fraction above 1 is ever reported."""
    sums = {k: sum(st.get(k, 0.0) for s in stats for st in s) for k in ("trace_ms", "shade_ms", "bounce_ms", "tail_ms", "paths_ms", "settled_ms")}
    tr_bytes = sh_bytes = bo_bytes = pa_bytes = se_bytes = 0.0
    for f, st_frame in zip(frames, zip(*stats)):
        # per-ray node fetches / primitive tests of this frame: one instrumented pass at 1 spp (untimed)
        f.ctx.profile(events=False, count_visits=True)
        st = f.render(spp=1, exchange=False)  # rank 0 alone: no collective here
        f.ctx.profile(events=True, count_visits=False)
        nodes_pr = st["node_visits"] / max(st["rays"], 1)
        prims_pr = st["prim_tests"] / max(st["rays"], 1)
        f.nodes_per_ray, f.prims_per_ray = nodes_pr, prims_pr
        # rays the tail kernel finished (the last few live paths of a batch) are timed apart from the per-bounce kernels
        f_rays = sum(s["rays"] - s["tail_rays"] for s in st_frame)
        f_paths = sum(s["paths"] for s in st_frame)
        alive = max(0.0, 1.0 - f_paths / max(f_rays, 1))  # every path ends exactly once
        mode = st_frame[0]["pipeline"]  # 0 split, 1 fused, 2 hybrid: camera rays fused, scattered rays split, 3 whole paths
        if mode == 3:
            # k_paths keeps a path in registers from its camera ray to its end: a pixel index in and a radiance out per
            # path, the scene once per XCD and launch -- that is all it has to move
            pa_bytes += f_paths * (4.0 + 16.0) + scene_bytes_per_launch(f.info) * sum(s["paths_launches"] for s in st_frame)
            continue
        fused_rays = f_rays if mode == 1 else (f_paths if mode == 2 else 0)
        split_rays = f_rays - fused_rays
        if fused_rays:
            primary = f_paths / max(fused_rays, 1)
            bo_bytes += fused_rays * bounce_bytes_per_ray(f.info, primary, alive, nodes_pr, prims_pr)
            bo_bytes += scene_bytes_per_launch(f.info) * sum(s["bounce_kernel_launches"] for s in st_frame)
        if split_rays:
            # (rays whose hit the bounce before them has settled are followed by k_settled in registers: neither the trace
            # kernel nor k_shade sees them; k_settled moves a queue entry in -- 56 B of state + 4 B of distance -- and,
            # for about as many paths, 56 B out when the path goes back to the tree)
            settled_rays = sum(s.get("settled_rays", 0) for s in st_frame)
            walked_rays = split_rays - settled_rays
            tr_bytes += walked_rays * trace_bytes_per_ray(f.info, nodes_pr, prims_pr)
            tr_bytes += scene_bytes_per_launch(f.info) * sum(s["trace_launches"] for s in st_frame)
            sh_bytes += walked_rays * shade_bytes_per_ray(alive)
            se_bytes += sum(s.get("settled_entries", 0) for s in st_frame) * (60.0 + 56.0)
    launches = {k: sum(st.get(k, 0) for s in stats for st in s) for k in ("trace_launches", "shade_launches", "bounce_kernel_launches", "paths_launches", "settled_launches")}
    cands = [("k_trace", sums["trace_ms"], tr_bytes, launches["trace_launches"]), ("k_shade", sums["shade_ms"], sh_bytes, launches["shade_launches"]),
             ("k_bounce", sums["bounce_ms"], bo_bytes, launches["bounce_kernel_launches"]), ("k_paths", sums["paths_ms"], pa_bytes, launches["paths_launches"]),
             ("k_settled", sums["settled_ms"], se_bytes, launches["settled_launches"])]

    def one(name, ms, nbytes, nl):
        t = measured_traffic(workload, name, width, spp) if use_committed else None
        traffic = round(t[0] * 1.0 / max(t[2], 1), 1) if t else None  # per launch of the profiled run
        per_launch = nbytes / max(nl, 1)
        source = "algorithmic"
        if ms > 0 and nbytes / (ms * 1e-3) / 1e9 > PEAK_HBM_GBPS and traffic:
            # more than HBM can deliver: the upper tree levels came from L2 / the Infinity Cache.  Charge what was measured.
            per_launch, source = traffic, "measured traffic (algorithmic node bytes exceed the HBM peak)"
        ach = per_launch * nl / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        if ach > PEAK_HBM_GBPS:
            # no measured traffic committed for this configuration: the figure is capped, not evidence
            source = f"algorithmic, capped at the HBM peak (uncapped {ach:.0f} GB/s: upper tree levels are served by L2 / Infinity Cache)"
            per_launch *= PEAK_HBM_GBPS / ach
            ach = PEAK_HBM_GBPS
        d = {"kernel": name, "achieved": round(ach, 2), "frac": round(ach / PEAK_HBM_GBPS, 5), "launches": int(nl),
             "avg_launch_ms": round(ms / max(nl, 1), 5), "algorithmic_bytes_per_launch": round(per_launch, 1), "bytes_source": source,
             "traffic": traffic, "traffic_source": t[1] if t else None}
        pmc = pmc_profile(workload, name, width, spp) if use_committed else None
        if pmc:
            d["valu"] = pmc
        return d

    ranked = sorted(cands, key=lambda k: -k[1])
    top = one(*ranked[0])
    # what limits the kernel: far below the HBM roofline with the VALUs busy means instruction issue, not memory
    valu = top.get("valu")
    bound = "valu" if (top["frac"] < 0.5 and (valu is None or valu["valu_busy"] >= 0.5)) else "hbm"
    roof = {"bound": bound, "kernel": top["kernel"], "achieved": top["achieved"], "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": top["frac"], "traffic": top["traffic"], "traffic_source": top["traffic_source"], "launches": top["launches"],
            "avg_launch_ms": top["avg_launch_ms"], "algorithmic_bytes_per_launch": top["algorithmic_bytes_per_launch"],
            "bytes_source": top["bytes_source"],
            # True: the scene fits an XCD's L2, so per-ray node/primitive fetches are cache hits and only the
            # path-state stream (plus one scene read per XCD and launch) is charged to HBM
            "scene_resident_in_l2": all(f.info["device_bytes"] <= L2_PER_XCD for f in frames),
            "trace_ms_per_step": round(sums["trace_ms"] / steps, 3), "shade_ms_per_step": round(sums["shade_ms"] / steps, 3),
            "bounce_ms_per_step": round(sums["bounce_ms"] / steps, 3), "tail_ms_per_step": round(sums["tail_ms"] / steps, 3),
            "paths_ms_per_step": round(sums["paths_ms"] / steps, 3), "settled_ms_per_step": round(sums["settled_ms"] / steps, 3),
            "nodes_per_ray": [round(f.nodes_per_ray, 2) for f in frames], "prims_per_ray": [round(f.prims_per_ray, 2) for f in frames]}
    # The two whole-step figures of the survey's measurement plan, over ALL kernels and the step's wall time: every ray
    # charged 32 B per node visit + 32 B per primitive test + 128 B of path state (served by L1/L2 for a resident scene, so
    # not an HBM figure), and the compulsory part alone, 128 B of path state per ray.
    step_ms = sum(st["render_ms"] for s in stats for st in s)
    if step_ms > 0:
        b_ray = compulsory = 0.0
        for f, st_frame in zip(frames, zip(*stats)):
            r = sum(s["rays"] for s in st_frame)
            b_ray += r * (32.0 * f.nodes_per_ray + 32.0 * f.prims_per_ray + 128.0)
            compulsory += r * 128.0
        roof["survey_b_ray_frac"] = round(b_ray / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)
        roof["compulsory_128B_frac"] = round(compulsory / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)
        roof["survey_note"] = ("survey_b_ray_frac charges every node visit and primitive test as if it came from HBM; for a scene resident "
                               "in L2 those are cache hits, so it is a cache-bandwidth figure and may exceed 1; compulsory_128B_frac is the "
                               "part that has to touch HBM in a wavefront formulation")
    if valu:
        roof["valu"] = valu
    if bound == "valu":
        roof["note"] = ("achieved/peak/frac are the kernel's algorithmic HBM bytes against the HBM peak; the kernel is bound by "
                        "VALU instruction issue under divergence (valu: busy share of SIMD cycles, lanes active per instruction) "
                        "and, where valu.vmem_lane_loads_per_cu_clk is near 1, by the address cycles of its scattered loads (one "
                        "compute-unit clock per active lane and load instruction), not by memory bandwidth")
    # the other kernels that take a real share of the step, in the same terms
    others = [one(*k) for k in ranked[1:] if k[1] > 0.15 * ranked[0][1] and k[3]]
    if others:
        roof["other_kernels"] = others
    return roof


def other_configs(device, n_synth):
    """One step each of BASELINE.json's five configs at their full sizes, outside the timed steps of the default
    workload: ms per frame, Mrays/s, the kernel that took the most time and its HBM figures.  A config is warmed up with
    one full render when that is cheap (under 2^31 paths) and with a one-sample render of the same frames otherwise
    (allocations, kernel loading); the frames of a config are rendered one after the other (a context holds up to 78 GB
    of path queues; the two of the largest config fit side by side)."""
    rows = []
    for wl in OTHER_CONFIGS:
        frames_cfg, width, aspect, spp = WORKLOADS[wl]
        frames, st, dt, build_s = [], [], 0.0, 0.0
        for n in frames_cfg:
            t0 = time.perf_counter()
            f = Frame(n, width, aspect, spp, 0, 1, device, n_synth)
            build_s += time.perf_counter() - t0
            f.ctx.profile(events=True)
            n_paths = f.w * f.h * spp
            f.ctx.reserve(n_paths)
            f.render(spp=spp if n_paths <= (1 << 31) else 1, exchange=False)  # allocations, kernel loading
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            last = f.render(exchange=False)
            torch.cuda.synchronize()
            one = time.perf_counter() - t0
            if one < 0.1:  # a frame of a few milliseconds: the mean of ten renders instead of one sample
                t0 = time.perf_counter()
                for _ in range(10):
                    last = f.render(exchange=False)
                torch.cuda.synchronize()
                one = (time.perf_counter() - t0) / 10
            st.append(last)
            dt += one
            frames.append(f)
        roof = kernel_rooflines(frames, [st], 1, wl, use_committed=True)
        rays = sum(s["rays"] for s in st)
        rows.append({"workload": wl, "frames": list(frames_cfg), "width": width, "height": frames[0].h, "spp": spp,
                     "ms_per_frame": round(dt * 1e3 / len(frames), 3), "ms_per_step": round(dt * 1e3, 3),
                     "mrays_per_s": round(rays / dt / 1e6, 1), "rays_per_step": rays, "scene_build_s": round(build_s, 2),
                     "dominant_kernel": roof["kernel"], "bound": roof["bound"], "hbm_frac": roof["frac"],
                     "hbm_bytes_source": roof["bytes_source"], "traffic_source": roof["traffic_source"],
                     "survey_b_ray_frac": roof.get("survey_b_ray_frac"), "compulsory_128B_frac": roof.get("compulsory_128B_frac"),
                     "kernel_ms": {k[:-12]: roof[k] for k in roof if k.endswith("_ms_per_step") and roof[k] > 0}})
        for f in frames:
            f.ctx.close()
        del frames
        torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (changes the workload; recorded in config)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--synth-n", type=int, default=10_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0, help="CPU work the cpu_baseline sample is sized to")
    ap.add_argument("--no-others", action="store_true", help="skip the one-frame renders of the other BASELINE configs (config.others)")
    ap.add_argument("--no-profile", action="store_true", help="skip HIP-event kernel timing (roofline becomes null)")
    ap.add_argument("--frames-at-once", type=int, default=-1, choices=[-1, 0, 1],
                    help="render the frames of a step concurrently, one HIP stream and host thread each (default: only "
                         "when the step is split over several ranks, where one rank's share of a frame no longer fills the GPU)")
    ap.add_argument("--exchange", default="abi", choices=["abi", "torch"],
                    help="frame exchange for N > 1: abi = rt_gather_frame (RCCL communicator made and driven through the C ABI), "
                         "torch = torch.distributed all_gather_into_tensor; abi falls back to torch when RCCL cannot be loaded")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (exchange staged through host memory, ranks may share a GPU) "
                         "only rehearses the multi-rank path on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            torch.distributed.init_process_group("gloo")
    if args.gpus != world and rank == 0:
        log(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")

    frames_cfg, width, aspect, spp = WORKLOADS[args.workload]
    width = args.width or width
    spp = args.spp or spp
    # A rank's share of a frame is small when the step is split N ways: the tail of every launch and the host round
    # trips of one frame then leave the GPU partly idle.  The frames of a step are independent, so a rank renders them
    # at once, each on its own stream from its own host thread.  With one rank every launch fills the GPU anyway and
    # the frames run one after the other, which also keeps the per-kernel HIP-event times free of overlap.
    at_once = len(frames_cfg) > 1 and (args.frames_at_once == 1 or (args.frames_at_once < 0 and world > 1))
    frames = [Frame(n, width, aspect, spp, rank, world, local_rank, args.synth_n, gather_on_cpu=args.backend == "gloo",
                    own_stream=at_once, exchange=args.exchange)
              for n in frames_cfg]
    pool = None
    if at_once:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=len(frames))
    for f in frames:
        f.ctx.profile(events=not args.no_profile)
        # set-up, not work: path queues allocated, pixel list uploaded and kernels loaded before the timed region,
        # whatever --warmup says (one untimed pass of this rank's tiles; every step renders the frame again from zero).
        # Full-size unless the frame is huge, so that a kernel-trace average over the whole process compares with
        # the per-launch times of the timed steps.
        n_mine = len(rtamd.owned_pixels(f.w, f.h, f.tile, rank, world))
        f.ctx.reserve(n_mine * spp)
        f.render(spp=spp if n_mine * spp <= (1 << 30) else 1, exchange=False)
    if world > 1 and args.backend == "nccl":
        # first collective builds the RCCL communicator: keep that out of the timed steps as well
        torch.distributed.all_reduce(torch.zeros(1, device=f"cuda:{local_rank}"))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step():
        if pool is None:
            return [f.render() for f in frames]
        for f in frames:
            f.clear()
        torch.cuda.current_stream().synchronize()  # the buffers are zero before any context's stream adds to them
        st = list(pool.map(lambda f: f.trace(), frames))  # rt_render returns with its stream drained
        for f in frames:
            f.exchange()
        return st

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    barrier()
    elapsed = time.perf_counter() - t0

    my_rays = sum(st["rays"] for s in stats for st in s)
    my_paths = sum(st["paths"] for s in stats for st in s)
    t = torch.tensor([elapsed, float(my_rays), float(my_paths)], dtype=torch.float64,
                     device="cpu" if args.backend == "gloo" else f"cuda:{local_rank}")
    if world > 1:
        tmax = t[:1].clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        tsum = t[1:].clone()
        torch.distributed.all_reduce(tsum, op=torch.distributed.ReduceOp.SUM)
        elapsed, rays, paths = float(tmax[0]), float(tsum[0]), float(tsum[1])
    else:
        rays, paths = float(my_rays), float(my_paths)

    out = None
    if rank == 0:
        # dominant kernel and its roofline, from HIP events recorded around every launch of the timed steps
        roof = None
        if not args.no_profile:
            roof = kernel_rooflines(frames, stats, args.steps, args.workload, use_committed=True, width=args.width, spp=args.spp)
            if at_once:
                # the frames share the GPU, so a kernel's HIP-event time includes the time it waited for the others'
                # waves: the per-kernel figures of such a run are upper bounds of the time, lower bounds of the rate
                roof["overlapped_streams"] = True

        out = {
            "metric": "Mrays/s (paths x bounces)", "value": round(rays / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "frames": list(frames_cfg), "width": width, "height": frames[0].h,
                       "spp": spp, "max_depth": MAX_DEPTH, "tile_split": f"{frames[0].tile}x{frames[0].tile} tiles round-robin over {world} rank(s)",
                       "frames_at_once": bool(at_once), "exchange": frames[0].exchange_kind, "bvh_width": [f.info.get("bvh_width", 2) for f in frames],
                       "rays_per_step": rays / args.steps, "paths_per_step": paths / args.steps,
                       "sec_per_frame": round(elapsed / args.steps / len(frames), 5)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_cfg, width, aspect, args.cpu_baseline_seconds)
            # vs_baseline stays null (the reference publishes no number); this is the ratio to the CPU port timed above
            out["vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        if world == 1 and args.workload == DEFAULT_WORKLOAD and not (args.no_others or args.no_profile or args.spp or args.width):
            for f in frames:  # their queues (up to 72 GB) are not needed any more
                f.ctx.close()
            frames.clear()
            torch.cuda.empty_cache()
            out["config"]["others"] = other_configs(local_rank, args.synth_n)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
