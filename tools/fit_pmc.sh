#!/bin/bash
# SQ counters per kernel of the lean fit (GPU box): two --pmc passes of one fit_time.py command, averages per dispatch.
# usage: bash tools/fit_pmc.sh <tag> <fit_time.py arguments...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
d=gpurun_out/fit_pmc_$tag
mkdir -p $d
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $PWD/$d/a -- python3 tools/fit_time.py "$@" > /dev/null 2> $d/err_a.txt || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $PWD/$d/b -- python3 tools/fit_time.py "$@" > /dev/null 2> $d/err_b.txt || exit 1
python3 tools/pmc_summary.py $d/a > $d/summary.txt
python3 tools/pmc_summary.py $d/b >> $d/summary.txt
grep -c . $d/summary.txt
