#!/bin/bash
# round 5, run g: inter RQT on the device (fixtures' bytes, config 5 at 1080p / command line), per-XCD heavy-first order (me_int time + HBM traffic)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/g; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -q -m gpu -k "rqt or part or hiermr" 2>&1 | tail -8 > $O/pytest_g.txt
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_rc.py -q -m gpu -k "config5" 2>&1 | tail -12 >> $O/pytest_g.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hot1 -o kt -- python $R/bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt_hot1/kt_results.db > $O/kernel_stats_hot_1stream.txt; rm -rf $O/kt_hot1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/hbm_traffic.py $(ls $O/pmc_f/*counter_collection.csv | head -1) $(ls $O/pmc_w/*counter_collection.csv | head -1) 3840x2160 $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1; rm -rf $O/pmc_f $O/pmc_w
cd $R
cat $O/pytest_g.txt; head -14 $O/kernel_stats_hot_1stream.txt | cut -c1-150; head -4 $O/hbm_traffic.txt
