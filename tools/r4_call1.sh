#!/bin/bash
# one-off (r04): head A/B + stamps, fit block-cap experiment
cd "$GRAFT_REPO_ROOT"
bash tools/ab_so.sh build/libofx_v1.so build/libofx_v2.so > gpurun_out/r4_ab_head.txt 2>&1
cp ofighters_amd/libofx.so /tmp/keep_main.so
for mb in 1024 4096; do
  cp build/libofx_mb$mb.so ofighters_amd/libofx.so
  echo "== OFX_FIT_MAX_BLOCKS=$mb" >> gpurun_out/r4_fit_mb.txt
  timeout -k 10 300 python tools/fit_bisect.py 2048 --whole >> gpurun_out/r4_fit_mb.txt 2>&1
done
cp /tmp/keep_main.so ofighters_amd/libofx.so
echo "== default (2048)" >> gpurun_out/r4_fit_mb.txt
timeout -k 10 300 python tools/fit_bisect.py 2048 --whole >> gpurun_out/r4_fit_mb.txt 2>&1
bash tools/head_stamps.sh r4v2 > gpurun_out/r4_stamps.txt 2>&1
echo done
