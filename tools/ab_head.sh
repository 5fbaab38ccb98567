#!/bin/bash
# A/B of k_head_stream build variants on the GPU box: tools/ab_head.sh "<HEAD_EXTRA flags 1>" "<flags 2>" ...
# prints frames + stream HIP-event ms per variant (bench roofline.avg_kernel_ms) and the tick; restores the default build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp ofighters_amd/libofx.so /tmp/libofx_keep.so
trap 'cp /tmp/libofx_keep.so ofighters_amd/libofx.so; touch ofighters_amd/csrc/ofx_head.hip' EXIT   # the default build comes back whatever happens
for flags in "$@"; do
  make -C ofighters_amd/csrc HEAD_EXTRA="$flags" -B ofx_head.o >/dev/null 2>&1 && make -C ofighters_amd/csrc >/dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$flags]', 'head ms', round(d['roofline']['avg_kernel_ms'],3), 'tick ms', round(d['ms_per_step'],3))"
done
make -C ofighters_amd/csrc -B ofx_head.o >/dev/null 2>&1; make -C ofighters_amd/csrc >/dev/null 2>&1
