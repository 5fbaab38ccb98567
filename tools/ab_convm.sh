# tile-shape A/B of the MFMA trunk convolutions (diagnostic): OFX_CM_VARIANT 0..3
export TMPDIR=/tmp
mkdir -p gpurun_out/cm
for v in ${1:-0 1 2 3}; do
  OFX_CM_VARIANT=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/cm/v$v -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/cm/err$v.txt
  echo "variant $v"; python3 tools/kstats.py gpurun_out/cm/v$v 8 | grep convm
done
