#!/bin/bash
# one GPU call: policy parity tests, a short bench line and the per-kernel times (rocprofv3 --kernel-trace --stats)
# usage: tools/gpu_check.sh <tag>     (run on the GPU box through gpurun, from the repo root)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_policy.py -x -q -m gpu > gpurun_out/t_$tag.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/t_$tag.log
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/b_$tag.log 2>&1 && \
python - <<PY
import json
d=json.loads(open("gpurun_out/b_$tag.log").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"])
PY
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/bp_$tag.log 2>&1
python tools/kstats.py gpurun_out/prof_$tag 8
