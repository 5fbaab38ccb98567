export TMPDIR=/tmp
mkdir -p gpurun_out/c1
for v in 10 20 40 80; do
  OFX_C1_TH=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/c1/v$v -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/c1/err$v.txt
  echo "TH $v"; python3 tools/kstats.py gpurun_out/c1/v$v 8 | grep conv1_lut
done
