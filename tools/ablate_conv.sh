# stage ablation of the trunk conv kernels (diagnostic): OFX_CONV_ABLATE bit 1 = no global loads, 2 = no FMAs, 4 = no stores
export TMPDIR=/tmp
for a in 0 1 2 4 3 7; do
  OFX_CONV_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/ablc/a$a -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/ablc/err$a.txt
  echo "ablate=$a conv2/3: $(grep 'k_conv<8, 8, 10, 100' gpurun_out/ablc/a$a/*/*kernel_stats.csv | cut -d, -f9-11 | cut -c1-60)"
done
