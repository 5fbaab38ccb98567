"""Joins the FETCH_SIZE / WRITE_SIZE averages (tools/pmc_summary.py output) with the kernel durations (tools/kstats.py output) of
the same fit run into a per-kernel table: calls per replay, ms, GB read (x2 on gfx950) / written, TB/s.
usage: python tools/fit_hbm_table.py <pmc summary> <kstats> <replays in the run>"""
import re, sys
hb, ks = {}, {}
for line in open(sys.argv[1]):
    name, rest = line.rstrip('\n').split('\t', 1)
    m = re.match(r'(FETCH_SIZE|WRITE_SIZE)=([0-9.e+\-]+)', rest)
    hb.setdefault(name.strip()[:58], {})[m.group(1)] = float(m.group(2))
for line in open(sys.argv[2]):
    m = re.match(r'(.*?)\s+(\d+)\s+([0-9.]+) ms(\s+[0-9.]+%)?$', line.rstrip())
    if m: ks[m.group(1).strip()[:58]] = (int(m.group(2)), float(m.group(3)))
reps = int(sys.argv[3])
rows = []
for k, v in hb.items():
    if k not in ks: continue
    calls, ms = ks[k]
    rd, wr = v.get('FETCH_SIZE', 0) * 1024 * 2, v.get('WRITE_SIZE', 0) * 1024
    rows.append((calls / reps * ms, k, calls // reps, ms, rd / 1e9, wr / 1e9, (rd + wr) / 1e9 / ms))
rows.sort(reverse=True)
print("%-60s %5s %9s %9s %9s %9s" % ("kernel", "calls", "ms", "read GB", "write GB", "TB/s"))
tot = totb = ftot = ftotb = 0
for t, k, c, ms, rd, wr, bw in rows:
    print("%-60s %5d %9.3f %9.2f %9.2f %9.2f" % (k, c, ms, rd, wr, bw)); tot += t; totb += (rd + wr) * c
    if "::f_" in k or k.startswith("t_"):       # ofx_fit.hip / ofx_train.hip: the fit itself (the rest: target forward, the
        ftot += t; ftotb += (rd + wr) * c        # tool's 12-tick rollout in front, counted as calls / replays)
print("sum over all kernels of the run per replay: %.1f ms, %.0f GB, %.2f TB/s" % (tot, totb, totb / tot))
print("sum over the fit's own kernels (f_*, t_*) per replay: %.1f ms, %.1f GB, %.2f TB/s" % (ftot, ftotb, ftotb / ftot))
