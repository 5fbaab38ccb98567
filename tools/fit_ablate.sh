#!/bin/bash
# kernel times of the fit under the diagnostic builds build/libofx_abl{0,1,2}.so (make FIT_EXTRA=-DOFX_FIT_ABLATE=n):
# 0 = product, 1 = forward tiles not filled, 2 = forward tiles not computed.  Results of 1 / 2 are wrong by construction.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp ofighters_amd/libofx.so /tmp/libofx_keep.so
trap 'cp /tmp/libofx_keep.so ofighters_amd/libofx.so' EXIT   # an interrupted run must not leave a variant installed
for a in 0 1 2; do
  cp build/libofx_abl$a.so ofighters_amd/libofx.so
  rm -rf gpurun_out/abl_stats
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_stats -- python3 tools/fit_time.py 2048 > /dev/null 2>&1
  echo "== OFX_FIT_ABLATE=$a"
  python3 tools/kstats.py gpurun_out/abl_stats 40 | grep "f_conv_fwd"
done
rm -rf gpurun_out/abl_stats
