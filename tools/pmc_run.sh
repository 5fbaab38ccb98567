# SQ counters of one lock-step (diagnostic).  Usage (GPU box): bash tools/pmc_run.sh <tag>
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $PWD/gpurun_out/pmc_$1/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> gpurun_out/pmc_$1/err_a.txt
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $PWD/gpurun_out/pmc_$1/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> gpurun_out/pmc_$1/err_b.txt
python3 tools/pmc_summary.py gpurun_out/pmc_$1/a > gpurun_out/pmc_$1/summary.txt
python3 tools/pmc_summary.py gpurun_out/pmc_$1/b >> gpurun_out/pmc_$1/summary.txt
