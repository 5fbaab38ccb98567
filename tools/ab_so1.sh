#!/bin/bash
# A/B of prebuilt library variants on the one-policy-ship line with the exact sparse trunk, inside ONE GPU call:
# tools/ab_so1.sh build/libofx_a.so build/libofx_b.so ...   (each twice, interleaved; restores the default build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp ofighters_amd/libofx.so /tmp/libofx_keep.so
trap 'cp /tmp/libofx_keep.so ofighters_amd/libofx.so' EXIT
for rep in 1 2; do
  for so in "$@"; do
    cp "$so" ofighters_amd/libofx.so
    timeout -k 10 200 python bench.py --steps 150 --warmup 30 --policy-ships 1 --trunk-sparse --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$so]', 'tick ms', round(d['ms_per_step'],3), 'k/s', round(d['value']/1e3,1))"
  done
done
