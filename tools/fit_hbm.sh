#!/bin/bash
# HBM traffic of the lean fit per kernel (GPU box): FETCH_SIZE and WRITE_SIZE in separate --pmc passes, durations from a
# --kernel-trace --stats pass of the same command, joined by tools/fit_hbm_table.py.
# usage: bash tools/fit_hbm.sh <tag> <fit_time.py arguments...>     e.g. bash tools/fit_hbm.sh ref2048 --reference 2048
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
d=gpurun_out/fit_hbm_$tag
mkdir -p $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $PWD/$d/t -- python3 tools/fit_time.py "$@" > $d/time.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $PWD/$d/f -- python3 tools/fit_time.py "$@" > /dev/null 2> $d/err_f.txt || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $PWD/$d/w -- python3 tools/fit_time.py "$@" > /dev/null 2> $d/err_w.txt || exit 1
python3 tools/kstats.py $d/t 80 > $d/kstats.txt
python3 tools/pmc_summary.py $d/f > $d/pmc.txt
python3 tools/pmc_summary.py $d/w >> $d/pmc.txt
grep fit_batch $d/time.log
python3 tools/fit_hbm_table.py $d/pmc.txt $d/kstats.txt 4 > $d/table.txt
tail -1 $d/table.txt
