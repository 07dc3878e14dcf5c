#!/bin/bash
# round 5, run l: timelines of the default-GOP encoder: 8 hardware queues without / with the anchor lane, 4 queues with the lane (normal priority)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/l; mkdir -p $O; cd /tmp
export TMPDIR=/tmp
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace -d $O/kt_$tag -o kt -- python $R/bench.py --leg encoded --hier-b 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$tag.json 2>/dev/null
  python $R/tools/rocpd_timeline.py $O/kt_$tag/kt_results.db 0.6 40 > $O/timeline_$tag.txt; rm -rf $O/kt_$tag
  head -1 $O/timeline_$tag.txt
}
run q8_nolane GPU_MAX_HW_QUEUES=8
run q8_lane GPU_MAX_HW_QUEUES=8 KS265_ANCHOR_LANE=1
run q4_lane KS265_ANCHOR_LANE=1
run q4_nolane_keyprio0 KS265_KEY_PRIO=0
