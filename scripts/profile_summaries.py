#!/usr/bin/env python3
"""Summaries of scripts/profile_round.sh's four rocprofv3 runs, written to <dir>/summary/ in the form profiles/ keeps:
  <tag>_summary.md                      per-kernel durations (kernel trace + stats)
  <tag>_<kernel>_hbm_traffic.json       FETCH_SIZE / WRITE_SIZE per kernel (read side doubled: gfx950 tallies a 128-byte
                                        read request as 64 bytes)
  <tag>_<kernel>_pmc.json               VALU busy share, lanes active per VALU instruction
Instrumented template instances (the bench's untimed counting pass) are left out."""
import collections, csv, glob, json, os, re, sys

d, tag, command = sys.argv[1], sys.argv[2], sys.argv[3]
out = os.path.join(d, "summary")
os.makedirs(out, exist_ok=True)


def find(sub, suffix):
    m = glob.glob(os.path.join(d, sub, "**", "p_" + suffix), recursive=True)
    return m[0] if m else None


def family(name):
    """k_trace / k_shade / ... of a timed (not instrumented) instance, else None."""
    m = re.search(r"(k_[a-z_]+)(<([^>]*)>)?", name)
    if not m:
        return None
    args = [a.strip() for a in (m.group(3) or "").split(",")] if m.group(3) else []
    if m.group(1) in ("k_trace", "k_bounce", "k_intersect_batch") and len(args) > 1 and args[1] == "true":
        return None
    return m.group(1)


# 1. durations
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarise_profile
stats = find("stats", "kernel_stats.csv")
if stats:
    summarise_profile.stats(stats, os.path.join(out, f"{tag}_summary.md"), f"rocprofv3 --kernel-trace --stats, {tag}", command)


def per_kernel(sub):
    """family -> (dispatches, counter sums, ns) of one pmc run"""
    cc, kt = find(sub, "counter_collection.csv"), find(sub, "kernel_trace.csv")
    if not cc:
        return {}
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt))} if kt else {}
    agg = collections.defaultdict(lambda: {"n": 0, "ns": 0, "c": collections.defaultdict(float)})
    seen = set()
    for r in csv.DictReader(open(cc)):
        f = family(r["Kernel_Name"])
        if not f:
            continue
        a = agg[f]
        a["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            a["n"] += 1
            t = trace.get(r["Dispatch_Id"])
            if t:
                a["ns"] += int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
    return agg


fetch, write, sq = per_kernel("fetch"), per_kernel("write"), per_kernel("sq")
for f in sorted(set(fetch) & set(write)):
    rb = 2.0 * fetch[f]["c"]["FETCH_SIZE"] * 1024.0
    wb = write[f]["c"]["WRITE_SIZE"] * 1024.0
    doc = {"kernel": f, "command": command, "dispatches_fetch_pass": fetch[f]["n"], "dispatches_write_pass": write[f]["n"],
           "FETCH_SIZE_KiB_sum": fetch[f]["c"]["FETCH_SIZE"], "WRITE_SIZE_KiB_sum": write[f]["c"]["WRITE_SIZE"],
           "read_bytes": rb, "write_bytes": wb, "bytes_per_launch": rb / max(fetch[f]["n"], 1) + wb / max(write[f]["n"], 1),
           "bytes_per_step": rb + wb, "kernel_ms_fetch_pass": fetch[f]["ns"] / 1e6,
           "note": "read = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B requests as 64 B); write = WRITE_SIZE x 1024; "
                   "bytes_per_step is the whole profiled run (all its steps)"}
    if fetch[f]["ns"]:
        doc["GBps_over_kernel_time"] = (rb + wb) / fetch[f]["ns"]
    json.dump(doc, open(os.path.join(out, f"{tag}_{f}_hbm_traffic.json"), "w"), indent=1)
for f, a in sorted(sq.items()):
    c = a["c"]
    if not c.get("SQ_ACTIVE_INST_VALU"):
        continue
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0)  # summed over the 8 XCDs by rocprofv3
    doc = {"kernel": f, "command": command, "dispatches": a["n"], "kernel_ms": a["ns"] / 1e6,
           "lane_use": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0),
           "counters": {k: v for k, v in sorted(c.items())}}
    if cyc:
        # 1024 SIMDs; ACTIVE_INST_VALU counts quad-cycles of VALU issue per SIMD
        doc["valu_busy"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (cyc / 8.0 * 1024.0)
        doc["clock_GHz"] = cyc / 8.0 / max(a["ns"], 1)
    json.dump(doc, open(os.path.join(out, f"{tag}_{f}_pmc.json"), "w"), indent=1)
    print(f, {k: round(v, 4) for k, v in doc.items() if isinstance(v, float)})
