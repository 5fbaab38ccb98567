export TMPDIR=/tmp
for a in 0 8; do
  OFX_HT_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/abl/a$a -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/abl/err$a.txt
  echo "ablate=$a $(grep k_head_tail gpurun_out/abl/a$a/*/*kernel_stats.csv | cut -d, -f4)"
done
