# NOTE: the hooks are compiled out of the production build: rebuild ofx_policy.o with -DOFX_ABLATE_HOOKS=1 first.
# Stage ablation of k_head_tail (diagnostic): OFX_HT_ABLATE bits 1 no stage-A loads, 2 no stage B, 4 no stage C,
# 8 no border passes.  Usage (GPU box): bash tools/ablate_head_tail.sh "0 8 2 4"
export TMPDIR=/tmp
mkdir -p gpurun_out/abl
for a in ${1:-0 8}; do
  OFX_HT_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/abl/a$a -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/abl/err$a.txt
  echo "ablate=$a $(grep k_head_tail gpurun_out/abl/a$a/*/*kernel_stats.csv | cut -d, -f4)"
done
