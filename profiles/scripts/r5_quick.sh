#!/bin/bash
# round-5 measurement loop in ONE gpurun call: GPU suite (or a subset: $1 = pytest args), kernel trace + SQ counters of the hot leg on one stream, the driver's bench line.
# usage:  gpurun --timeout 900 -- "bash tools/r5_quick.sh [TAG] [pytest args]"
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; shift; O=$R/gpurun_out/r05/$TAG; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "kernel_src_sha $(python -c 'from ks265codec_amd.build import source_sha; print(source_sha())')" > $O/stamp.txt
PYARGS=${@:-tests -q -m gpu}
timeout 1200 python -m pytest $PYARGS 2>&1 | tail -25 > $O/pytest_gpu.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hot1 -o kt -- python $R/bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt_hot1/kt_results.db > $O/kernel_stats_hot_1stream.txt; rm -rf $O/kt_hot1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/sq_summary.py $(ls $O/pmc_sq2/*counter_collection.csv | head -1) > $O/sq_counters.txt 2>&1; rm -rf $O/pmc_sq2
cd $R
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_default.err | grep '^{' | tail -1 > $O/bench_line_default.json
cat $O/pytest_gpu.txt | tail -8; head -24 $O/kernel_stats_hot_1stream.txt; head -16 $O/sq_counters.txt | cut -c1-220; head -c 1500 $O/bench_line_default.json; echo; tail -3 $O/bench_default.err
