"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run (csv or rocpd sqlite output): name / calls /
average ms (diagnostic helper)."""
import csv, glob, sqlite3, sys
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dbs = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)
if dbs:
    c = sqlite3.connect(dbs[0])
    for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit %d" % n):
        print("%-72s %5d %9.3f ms %5.1f%%" % (name[:72], calls, avg / 1e3, pct))
else:
    f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
    for r in list(csv.DictReader(open(f)))[:n]:
        print("%-72s %5s %9.3f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6))
