#!/bin/bash
# round 5, run i: timeline of the default-GOP encoder with / without the anchor lane
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/i; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_enc_api.py -q -m gpu -k "anchor_lane" 2>&1 | tail -4 > $O/pytest_i.txt
cd /tmp
for v in lane nolane; do
  [ $v = nolane ] && export KS265_NO_ANCHOR_LANE=1
  timeout 300 rocprofv3 --kernel-trace -d $O/kt_$v -o kt -- python $R/bench.py --leg encoded --hier-b 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$v.json 2>/dev/null
  python $R/tools/rocpd_timeline.py $O/kt_$v/kt_results.db 0.6 40 > $O/timeline_$v.txt
  rm -rf $O/kt_$v
done
cd $R; cat $O/pytest_i.txt; head -1 $O/timeline_lane.txt; head -1 $O/timeline_nolane.txt
