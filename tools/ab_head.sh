#!/bin/bash
# A/B of k_head_stream build variants / ablations on the GPU box: tools/ab_head.sh "<HEAD_EXTRA flags>" "<env list>"
# prints frames+stream HIP-event ms per variant (bench roofline.avg_kernel_ms) and the tick
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
flags="$1"; shift
make -C ofighters_amd/csrc HEAD_EXTRA="$flags" -B ofx_head.o >/dev/null 2>&1 && make -C ofighters_amd/csrc >/dev/null 2>&1 || { echo build failed; exit 1; }
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', 'head ms', round(d['roofline']['avg_kernel_ms'],3), 'tick ms', round(d['ms_per_step'],3))"
done
