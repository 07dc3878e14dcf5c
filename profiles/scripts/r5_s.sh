#!/bin/bash
# round 5, run s/t: kernel changes at the end of the round (phase H shared, candidates gate list): parity + kernel times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/${RUN:-s}; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_frame.py tests/test_gpu_golden.py tests/test_gpu_stream.py -q -m gpu -x 2>&1 | tail -6 > $O/pytest_s.txt
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -6 >> $O/pytest_s.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hot1 -o kt -- python $R/bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline > $O/bench_hot1.json 2>/dev/null
python $R/tools/rocpd_stats.py $O/kt_hot1/kt_results.db > $O/kernel_stats_hot_1stream.txt; rm -rf $O/kt_hot1
cd $R; cat $O/pytest_s.txt; head -6 $O/kernel_stats_hot_1stream.txt | cut -c1-150
