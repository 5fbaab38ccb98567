"""Print the load / wait / MFMA pattern of the basic blocks of a kernel that hold many MFMAs (from a hipcc -S file)."""
import re, sys
path, kernel = sys.argv[1], sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 30
s = open(path).read()
a = s.index(kernel + ':')
b = s.index('.Lfunc_end', a)
blocks, cur = [], ['entry', []]
for l in s[a:b].split('\n'):
    t = l.split(';')[0].strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur); cur = [m.group(1), []]; continue
    if not t or t.startswith('.') or re.match(r'^[\w.$]+:', t): continue
    cur[1].append(t)
blocks.append(cur)
for name, ins in blocks:
    if sum(i.startswith('v_mfma') for i in ins) < minm: continue
    print('==', name, len(ins))
    run = 0
    for t in ins:
        if t.startswith('v_mfma'): run += 1; continue
        if run: print(f'      mfma x{run}'); run = 0
        print('  ', t[:72])
    if run: print(f'      mfma x{run}')
