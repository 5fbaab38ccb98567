#!/bin/bash
# Diagnostic: the fit's parity tests for several prebuilt library variants in ONE GPU call:
# tools/ab_fit_test.sh build/libofx_a.so build/libofx_b.so ...   (restores the default build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp ofighters_amd/libofx.so /tmp/libofx_keep.so
trap 'cp /tmp/libofx_keep.so ofighters_amd/libofx.so' EXIT
for so in "$@"; do
  cp "$so" ofighters_amd/libofx.so
  echo "== $so"
  timeout -k 10 400 python -m pytest tests/test_train.py -q -m gpu -k "${FIT_TESTS:-lean}" 2>&1 | grep -E "passed|failed|FAILED|conv1\.|AssertionError" | cut -c1-220
done
