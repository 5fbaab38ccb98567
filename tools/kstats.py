"""Print a rocprofv3 kernel_stats.csv as name / calls / average ms (diagnostic helper)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-72s %5s %9.3f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6))
