# stage ablation of the conv kernels (diagnostic): OFX_CONV_ABLATE bit 1 = no global loads, 2 = no FMAs, 4 = no stores
export TMPDIR=/tmp
mkdir -p gpurun_out/ablc
for a in 0 1 2 4 7; do
  OFX_CONV_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/ablc/a$a -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/ablc/err$a.txt
  python3 - "$a" <<'PY'
import csv, glob, sys
a = sys.argv[1]
for f in glob.glob("gpurun_out/ablc/a%s/*/*kernel_stats.csv" % a):
    rows = {r["Name"]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(f))}
    print("ablate=%s" % a, {k[:34]: round(v, 3) for k, v in rows.items() if "k_conv" in k})
PY
done
