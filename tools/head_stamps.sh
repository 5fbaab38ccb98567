#!/bin/bash
# s_memtime stamps + stage ablations of k_head_stream on the GPU box (diagnostic build -DOFX_HEAD_HOOKS=1; restores the
# default build).  usage: tools/head_stamps.sh <tag> ["<extra HEAD_EXTRA flags>"]
tag=${1:-x}
extra=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp ofighters_amd/libofx.so /tmp/libofx_keep.so
trap 'cp /tmp/libofx_keep.so ofighters_amd/libofx.so; touch ofighters_amd/csrc/ofx_head.hip' EXIT   # the default build comes back whatever happens
mkdir -p gpurun_out
make -C ofighters_amd/csrc HEAD_EXTRA="-DOFX_HEAD_HOOKS=1 $extra" -B ofx_head.o >/dev/null 2>&1 && make -C ofighters_amd/csrc >/dev/null 2>&1 || { echo "build failed"; exit 1; }
run() { timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra 2>>gpurun_out/stamps_$tag.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'head ms', round(d['roofline']['avg_kernel_ms'],3), 'tick ms', round(d['ms_per_step'],3))"; }
: > gpurun_out/stamps_$tag.txt
OFX_HEAD_STAMPS=1 run "hooks build, stamps"
OFX_HEAD_ABLATE=1 run "no stage C (producers + stage A alone)"
OFX_HEAD_ABLATE=2 run "no stage B (consumers alone)"
make -C ofighters_amd/csrc -B ofx_head.o >/dev/null 2>&1; make -C ofighters_amd/csrc >/dev/null 2>&1
cat gpurun_out/stamps_$tag.txt | grep "head stamps"
