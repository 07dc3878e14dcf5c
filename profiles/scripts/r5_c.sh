#!/bin/bash
# round 5, run c: rdoQuant on the device (new operator), heavy-first dispatch of the integer search (parity + timeline + kernel trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/c; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rdoq.py -q -m gpu 2>&1 | tail -30 > $O/pytest_rdoq.txt
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_frame.py -q -m gpu -x 2>&1 | tail -6 > $O/pytest_c.txt
timeout 300 python tools/me_trace.py build/variants/libks265hip_trace.so > $O/me_trace.txt 2>&1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hot1 -o kt -- python $R/bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline > $O/hot1.json 2>/dev/null
python $R/tools/rocpd_stats.py $O/kt_hot1/kt_results.db > $O/kernel_stats_hot_1stream.txt; rm -rf $O/kt_hot1
KS265_ME_ORDER_OFF=1 timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hot1 -o kt -- python $R/bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline > $O/hot1_noorder.json 2>/dev/null
python $R/tools/rocpd_stats.py $O/kt_hot1/kt_results.db > $O/kernel_stats_hot_1stream_noorder.txt; rm -rf $O/kt_hot1
cd $R
cat $O/pytest_rdoq.txt $O/pytest_c.txt $O/me_trace.txt; head -8 $O/kernel_stats_hot_1stream.txt | cut -c1-150; head -8 $O/kernel_stats_hot_1stream_noorder.txt | cut -c1-150
