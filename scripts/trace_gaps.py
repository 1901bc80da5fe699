"""Per-dispatch durations and idle gaps from a rocprofv3 --kernel-trace SQLite output (rocpd).
   python scripts/trace_gaps.py <results.db> [last_n_dispatches_to_list]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = {r[0].rsplit('_00000000', 1)[0]: r[0] for r in db.execute("select name from sqlite_master where type='table'")}
kd, ks = tabs['rocpd_kernel_dispatch'], tabs['rocpd_info_kernel_symbol']
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
rows = db.execute(f"select d.start, d.end, s.kernel_name, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n[:n.find('(')] if '(' in n else n[:60]
# split into renders at every k_accumulate
seg, segs = [], []
for r in rows:
    seg.append(r)
    if 'k_accumulate' in r[2]:
        segs.append(seg); seg = []
for i, s in enumerate(segs):
    s = [r for r in s if 'k_' in r[2]]
    busy = sum(r[1] - r[0] for r in s)
    span = s[-1][1] - s[0][0]
    per = {}
    for r in s:
        k = short(r[2]); per.setdefault(k, [0, 0]); per[k][0] += 1; per[k][1] += r[1] - r[0]
    print(f"render {i}: {len(s)} dispatches, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:40s} x{v[0]:3d} {v[1]/1e6:8.3f} ms")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if n and segs:
    s = [r for r in segs[-1] if 'k_' in r[2]]
    prev = None
    for r in s[:n]:
        gap = (r[0] - prev) / 1e3 if prev else 0
        print(f"  gap {gap:7.1f} us  dur {(r[1]-r[0])/1e3:8.1f} us  grid {r[3]:8d}  {short(r[2])}")
        prev = r[1]
