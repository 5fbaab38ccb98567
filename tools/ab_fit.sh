#!/bin/bash
# A/B of prebuilt library variants on the fit inside ONE GPU call: tools/ab_fit.sh ROWS build/libofx_a.so build/libofx_b.so ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rows=$1; shift
cp ofighters_amd/libofx.so /tmp/libofx_keep.so
trap 'cp /tmp/libofx_keep.so ofighters_amd/libofx.so' EXIT   # an interrupted run must not leave a variant installed
for rep in 1 2; do
  for so in "$@"; do
    cp "$so" ofighters_amd/libofx.so
    echo "[$so] $(timeout -k 10 200 python tools/fit_time.py $rows 2>&1 | tail -1 | cut -c1-80)"
  done
done
