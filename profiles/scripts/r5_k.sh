#!/bin/bash
# round 5, run k: anchor lane waiting for its upload directly; stream priority 1 / 0; 4 / 8 hardware queues
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/k; mkdir -p $O; cd $R
export TMPDIR=/tmp
for q in 4 8; do for p in 1 0; do
  GPU_MAX_HW_QUEUES=$q KS265_ANC_PRIO=$p timeout 300 python bench.py --leg encoded --hier-b 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_q${q}_p$p.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/bench_q${q}_p$p.json").read().strip().splitlines()[-1])
print("q$q prio $p hier", d["value"])
PY
done; done | tee $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --leg encoded --hier-b 8 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $O/kt/kt_results.db 0.6 40 > $O/timeline_lane.txt; rm -rf $O/kt
head -1 $O/timeline_lane.txt
