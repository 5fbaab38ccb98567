# compiler flag A/B for ofx_policy.hip (diagnostic; rebuilds libofx.so on the GPU box, one variant per line of the list)
export TMPDIR=/tmp
BASE="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -fPIC"
i=0
while IFS= read -r extra; do
  i=$((i+1))
  (cd ofighters_amd/csrc && hipcc $BASE $extra -c ofx_policy.hip -o ofx_policy.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libofx.so ofx_api.o ofx_step.o ofx_raster.o ofx_nn.o ofx_policy.o ofx_replay.o) || { echo "variant $i [$extra]: build failed"; continue; }
  echo "variant $i [$extra]"
  timeout -k 10 300 python -m pytest tests/test_gpu_policy.py -x -q -m gpu 2>&1 | tail -1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/sched/v$i -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/sched_err.txt
  python3 tools/kstats.py gpurun_out/sched/v$i 6 | grep "head_tail\|convm<8, 2\|conv1"
done <<'LIST'
-mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -enable-post-misched=0
-mllvm -amdgpu-sched-strategy=iterative-maxocc -O2
-mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -amdgpu-schedule-relaxed-occupancy=true
-mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -amdgpu-mfma-vgpr-form=true
-mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -amdgpu-use-amdgpu-trackers=1
LIST
