#!/bin/bash
# round 5, run h: the anchor lane (stream identical with the lane off; default-GOP bench with / without it), the order kernel by bisection (me_int parity + time)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/h; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_enc_api.py -q -m gpu -k "anchor_lane or graphs or lanes" 2>&1 | tail -8 > $O/pytest_h.txt
timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_golden.py -q -m gpu -x 2>&1 | tail -5 >> $O/pytest_h.txt
timeout 900 python -m pytest tests/test_gpu_rc.py -q -m gpu -k "config5 or hier or default" 2>&1 | tail -8 >> $O/pytest_h.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_lane.json 2> $O/bench_lane.err
KS265_NO_ANCHOR_LANE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_nolane.json 2> $O/bench_nolane.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hot1 -o kt -- python $R/bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt_hot1/kt_results.db > $O/kernel_stats_hot_1stream.txt; rm -rf $O/kt_hot1
cd $R
cat $O/pytest_h.txt; tail -c 1500 $O/bench_lane.json; echo; tail -c 1500 $O/bench_nolane.json; echo; head -12 $O/kernel_stats_hot_1stream.txt | cut -c1-150
