"""Average rocprofv3 --pmc counters per dispatch and kernel (diagnostic helper).
usage: python3 tools/pmc_summary.py <rocprof output dir> [kernel-name substring]"""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k + "\t" + "\t".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
