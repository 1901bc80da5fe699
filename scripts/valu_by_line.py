"""Static VALU instruction count per source line for one kernel (hipcc -S -gline-tables-only output)."""
import re, collections, sys
asm, pattern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
lines = open(asm).read().split('\n')
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r'^_ZN.*%s.*:' % pattern, l):
        start = i
    elif start is not None and re.match(r'^\.Lfunc_end\d+:', l):
        end = i
        break
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
cnt = collections.Counter()
cur = None
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
        continue
    t = l.strip()
    if t.startswith('v_'):
        cnt[cur] += 1
print('total static VALU', sum(cnt.values()))
byfile = collections.Counter()
for (f, ln), c in cnt.items():
    byfile[f] += c
print(dict(byfile))
for (f, ln), c in sorted(cnt.items(), key=lambda x: -x[1])[:top]:
    print(f, ln, c)
