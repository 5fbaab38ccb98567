#!/bin/bash
# A/B of prebuilt library variants inside ONE GPU call (boxes differ by ~1 %: never compare across calls):
# tools/ab_so.sh build/libofx_a.so build/libofx_b.so ...   each is timed twice, interleaved; restores the default build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp ofighters_amd/libofx.so /tmp/libofx_keep.so
trap 'cp /tmp/libofx_keep.so ofighters_amd/libofx.so' EXIT   # an interrupted run must not leave a variant installed
for rep in 1 2; do
  for so in "$@"; do
    cp "$so" ofighters_amd/libofx.so
    timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$so]', 'head ms', round(d['roofline']['avg_kernel_ms'],3), 'tick ms', round(d['ms_per_step'],3))"
  done
done
