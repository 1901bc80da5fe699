"""Per-kernel counter sums and durations from a rocprofv3 --pmc ... --kernel-trace run (rocpd SQLite output).
   python scripts/pmc_db.py <results.db> [kernel-substring]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
needle = sys.argv[2] if len(sys.argv) > 2 else "k_"
tabs = {r[0].rsplit('_00000000', 1)[0]: r[0] for r in db.execute("select name from sqlite_master where type='table'")}
pe, ip, kd, ks = tabs['rocpd_pmc_event'], tabs['rocpd_info_pmc'], tabs['rocpd_kernel_dispatch'], tabs['rocpd_info_kernel_symbol']
def fam(n):
    m = re.search(r'(k_[a-z_]+)', n)
    return m.group(1) if m else n[:40]
dur = {}
for name, ns, cnt in db.execute(f"select s.kernel_name, sum(d.end-d.start), count(*) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name"):
    if needle in name:
        f = fam(name); d = dur.setdefault(f, [0, 0]); d[0] += ns; d[1] += cnt
agg = {}
q = f"""select s.kernel_name, p.name, sum(e.value) from {pe} e join {ip} p on e.pmc_id = p.id
        join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name"""
for name, ctr, v in db.execute(q):
    if needle in name:
        agg.setdefault(fam(name), {}).setdefault(ctr, 0.0)
        agg[fam(name)][ctr] += v
for f, (ns, cnt) in sorted(dur.items(), key=lambda kv: -kv[1][0]):
    c = agg.get(f, {})
    cyc_xcd = c.get('GRBM_GUI_ACTIVE', 0) / 8.0   # summed over the 8 XCDs
    line = [f"{f}: n={cnt} ms={ns/1e6:.3f} clk={cyc_xcd/max(ns,1):.2f}GHz"]
    simd_cycles = cyc_xcd * 1024.0
    if 'SQ_ACTIVE_INST_VALU' in c: line.append(f"VALU busy={4*c['SQ_ACTIVE_INST_VALU']/simd_cycles:.3f}")
    if 'SQ_ACTIVE_INST_LDS' in c: line.append(f"LDS busy={4*c['SQ_ACTIVE_INST_LDS']/simd_cycles:.3f}")
    if 'TA_TA_BUSY_sum' in c: line.append(f"TA busy={c['TA_TA_BUSY_sum']/(cyc_xcd*256):.3f}")
    if 'TCP_PENDING_STALL_CYCLES_sum' in c: line.append(f"TCP pending stall={c['TCP_PENDING_STALL_CYCLES_sum']/(cyc_xcd*256):.3f}")
    if 'SQ_WAVE_CYCLES' in c: line.append(f"waves/SIMD={4*c['SQ_WAVE_CYCLES']/simd_cycles:.2f}")
    if 'SQ_WAIT_INST_ANY' in c and 'SQ_WAVE_CYCLES' in c: line.append(f"wait frac={c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f}")
    print('  '.join(line))
    print('     ' + '  '.join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
