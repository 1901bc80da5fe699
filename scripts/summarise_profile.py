#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the small summaries kept under profiles/.

  summarise_profile.py stats  <dir>/<prefix>_kernel_stats.csv  <out.md>  "<title>" "<command>"
  summarise_profile.py traffic <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <kernel-substring> [steps]

`traffic` adds up FETCH_SIZE and WRITE_SIZE (KiB) over the dispatches whose kernel name contains the
substring and does not contain the instrumented template flag pattern given by --exclude. gfx950 tallies
128-byte read requests at 64 bytes, so the read side is doubled.
"""
import csv
import json
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    cut = name.find("(")
    if cut > 0 and not name.startswith("at::"):
        name = name[:cut]
    return name[:90]


def stats(path, out, title, command):
    rows = list(csv.DictReader(open(path)))
    lines = ["# %s" % title, "", "Command (on the MI355X box): `%s`" % command, "",
             "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    # all template instances of one kernel together (bench.py times them as one kernel), without the instrumented
    # instances (k_trace / k_bounce / k_intersect_batch with the second template argument true) that bench.py only runs in its untimed counting pass
    fams = {}
    for r in rows:
        name = short(r["Name"])
        if name.startswith("k_") and "<" in name:
            args = name[name.index("<") + 1:].rstrip(">").split(",")
            if name[:name.index("<")] in ("k_trace", "k_bounce", "k_intersect_batch") and len(args) > 1 and args[1].strip() == "true":
                continue
            f = fams.setdefault(name[:name.index("<")], [0, 0.0])
            f[0] += int(r["Calls"])
            f[1] += float(r["TotalDurationNs"])
    for name, (calls, ns) in sorted(fams.items(), key=lambda kv: -kv[1][1]):
        lines.append("| **%s, all timed instances** | %d | %.2f | %.2f | |" % (name, calls, ns / 1e6, ns / calls / 1e3))
    for r in rows:
        lines.append("| %s | %s | %.2f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                        float(r["AverageNs"]) / 1e3, r["Percentage"]))
    open(out, "w").write("\n".join(lines) + "\n")


def traffic(fetch_csv, write_csv, out, needle, steps=1, exclude=None):
    def total(path):
        n, s = 0, 0.0
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if needle in k and not (exclude and exclude in k):
                n += 1
                s += float(r["Counter_Value"])
        return n, s
    nf, fetch_kib = total(fetch_csv)
    nw, write_kib = total(write_csv)
    read_b = 2.0 * fetch_kib * 1024.0
    write_b = write_kib * 1024.0
    doc = {"kernel": needle, "exclude": exclude, "dispatches_fetch_pass": nf, "dispatches_write_pass": nw,
           "FETCH_SIZE_KiB_sum": fetch_kib, "WRITE_SIZE_KiB_sum": write_kib,
           "read_bytes": read_b, "write_bytes": write_b,
           "bytes_per_launch": (read_b / max(nf, 1)) + (write_b / max(nw, 1)),
           "steps_profiled": steps, "bytes_per_step": (read_b + write_b) / steps,
           "note": "read = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B requests as 64 B); write = WRITE_SIZE x 1024"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(*sys.argv[2:6])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]) if len(sys.argv) > 6 else 1)
