"""Summarise a rocprofv3 --pmc run (counter_collection.csv + kernel_trace.csv) per kernel family."""
import csv, collections, re, sys
d, prefix = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f'{d}/{prefix}_counter_collection.csv')))
trace = {r['Dispatch_Id']: r for r in csv.DictReader(open(f'{d}/{prefix}_kernel_trace.csv'))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = set()
def fam(n):
    m = re.search(r'(k_\w+)(<[^>]*>)?', n)
    if not m: return None
    return m.group(1) + (m.group(2) or '')
for r in rows:
    k = fam(r['Kernel_Name'])
    if not k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id'])
        t = trace.get(r['Dispatch_Id'])
        if t:
            agg[k]['_ns'] += int(t['End_Timestamp']) - int(t['Start_Timestamp'])
            agg[k]['_n'] += 1
for k, v in agg.items():
    ns = v['_ns']
    out = [f"{k}: n={int(v['_n'])} ms={ns/1e6:.3f}"]
    if 'GRBM_GUI_ACTIVE' in v and ns:
        out.append(f"clk={v['GRBM_GUI_ACTIVE']/ns:.2f}GHz")
    cyc = v.get('GRBM_GUI_ACTIVE', ns * 2.4)
    if 'SQ_ACTIVE_INST_VALU' in v:
        out.append(f"VALU busy={4*v['SQ_ACTIVE_INST_VALU']/(cyc*1024):.3f}")
    for c in sorted(v):
        if c.startswith('_'): continue
        out.append(f"{c}={v[c]:.4g}")
    print('  '.join(out))
