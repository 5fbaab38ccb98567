# A/B of k_head_tail stage-C scheduling variants (diagnostic; rebuilds libofx.so on the GPU box, one variant per line)
export TMPDIR=/tmp
BASE="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -fPIC -I../../include -mllvm -amdgpu-sched-strategy=iterative-maxocc -mllvm -amdgpu-mfma-vgpr-form=true"
i=0
while IFS= read -r extra; do
  i=$((i+1))
  (cd ofighters_amd/csrc && hipcc $BASE $extra -c ofx_policy.hip -o ofx_policy.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libofx.so ofx_api.o ofx_step.o ofx_raster.o ofx_nn.o ofx_policy.o ofx_replay.o ofx_train.o) || { echo "variant $i [$extra]: build failed"; continue; }
  echo "variant $i [$extra]"
  timeout -k 10 300 python -m pytest tests/test_gpu_policy.py -x -q -m gpu 2>&1 | tail -1
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/ab_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])"
done
