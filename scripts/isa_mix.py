"""Static instruction mix of kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only): scripts/isa_mix.py file.s pattern..."""
import re, sys
from collections import Counter
src = open(sys.argv[1]).read().split('\n')
pats = sys.argv[2:]
start = None
name = None
def classify(op):
    if op.endswith('_f64') or '_f64_' in op: return 'f64'
    if op.startswith(('v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_u64_u32')): return 'mul32'
    if op.startswith(('v_rcp', 'v_rsq', 'v_sqrt', 'v_log', 'v_exp', 'v_sin', 'v_cos')): return 'trans'
    if op.startswith('v_'): return 'valu'
    if op.startswith(('s_load', 's_buffer')): return 'sload'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): return 'vmem'
    if op.startswith('ds_'): return 'lds'
    return 'other'
c = None
for l in src:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        if c is not None: print(name[:70], sum(c.values()), dict(c))
        c = None
        if any(p in m.group(1) for p in pats):
            name, c = m.group(1), Counter()
        continue
    if c is not None and l.startswith('\t'):
        t = l.strip()
        if not t or t[0] in '.;': continue
        if t.startswith('s_endpgm'):
            print(name[:70], sum(c.values()), dict(c)); c = None; continue
        c[classify(t.split()[0])] += 1
