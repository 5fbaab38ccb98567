# compiler scheduling strategy A/B for ofx_policy.hip (diagnostic; rebuilds libofx.so on the GPU box)
export TMPDIR=/tmp
cd ofighters_amd/csrc
for st in ${1:-iterative-maxocc iterative-ilp}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -fPIC -mllvm -amdgpu-sched-strategy=$st -c ofx_policy.hip -o ofx_policy.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libofx.so ofx_api.o ofx_step.o ofx_raster.o ofx_nn.o ofx_policy.o ofx_replay.o
  cd ../..
  echo "strategy $st"
  timeout -k 10 300 python -m pytest tests/test_gpu_policy.py -x -q -m gpu 2>&1 | tail -1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/sched/$st -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/sched_err.txt
  python3 tools/kstats.py gpurun_out/sched/$st 6 | grep "head_tail\|convm\|conv1"
  cd ofighters_amd/csrc
done
