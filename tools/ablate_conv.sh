# NOTE: the hooks are compiled out of the production build: rebuild ofx_policy.o with -DOFX_ABLATE_HOOKS=1 first.
# Stage ablation of the trunk convolutions (diagnostic): OFX_CONV_ABLATE bits 1 no global loads, 2 no MFMAs / FMAs,
# 4 no stores.  Usage (GPU box): bash tools/ablate_conv.sh "0 1 2 4 7"
export TMPDIR=/tmp
mkdir -p gpurun_out/cabl
for a in ${1:-0 1 2 4 7}; do
  OFX_CONV_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/cabl/a$a -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/cabl/err$a.txt
  echo "ablate $a"; python3 tools/kstats.py gpurun_out/cabl/a$a 8 | grep "convm\|k_conv"
done
