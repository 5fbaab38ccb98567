# A/B of prebuilt library variants on k_head_frames / k_head_stream under rocprofv3 (kernel averages), inside ONE GPU call:
# tools/ab_frames.sh build/libofx_a.so build/libofx_b.so ...   (restores the default build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp ofighters_amd/libofx.so /tmp/libofx_keep.so
trap 'cp /tmp/libofx_keep.so ofighters_amd/libofx.so' EXIT
for so in "$@" "$@"; do
  cp $so ofighters_amd/libofx.so
  rm -rf gpurun_out/prof_ab; rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ab -- python3 bench.py --steps 60 --warmup 10 --no-extra --no-cpu-baseline > /dev/null 2>&1
  echo "[$so] $(python3 tools/kstats.py gpurun_out/prof_ab 4 | grep -E 'k_head_frames|k_head_stream' | tr -s ' ' | tr '\n' ' ')"
done
