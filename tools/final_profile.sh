# Collects the round's judged artefacts on the GPU box (scratch output under gpurun_out/final_<tag>):
#   bench.json (default bench), bench_under_rocprof.json + kernel_stats.csv, SQ / HBM counters per kernel.
# Usage: bash tools/final_profile.sh <tag>
export TMPDIR=/tmp
O=$PWD/gpurun_out/final_$1
mkdir -p $O
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 120 --warmup 20 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/sq_a.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq_b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/sq_b.err || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/hbm_r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/hbm_r.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/hbm_w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $O/hbm_w.err || exit 1
python3 tools/pmc_summary.py $O/sq_a > $O/pmc_sq.txt
python3 tools/pmc_summary.py $O/sq_b >> $O/pmc_sq.txt
python3 tools/pmc_summary.py $O/hbm_r > $O/pmc_hbm.txt
python3 tools/pmc_summary.py $O/hbm_w >> $O/pmc_hbm.txt
rm -rf $O/stats/*/*kernel_trace.csv $O/sq_a $O/sq_b $O/hbm_r $O/hbm_w
cat $O/bench.json | cut -c1-400
python3 tools/kstats.py $O/stats 12
grep k_head $O/pmc_hbm.txt
