#!/bin/bash
# round 5, run j: HIP streams share hardware queues (GPU_MAX_HW_QUEUES, default 4): the default-GOP encoder with 4 / 8 / 16, anchor lane on / off
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/j; mkdir -p $O; cd $R
export TMPDIR=/tmp
for q in 4 8 16; do for v in lane nolane; do
  unset KS265_NO_ANCHOR_LANE; [ $v = nolane ] && export KS265_NO_ANCHOR_LANE=1
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --leg encoded --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${v}_q$q.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/bench_${v}_q$q.json").read().strip().splitlines()[-1])
print("$v q$q hier", d["value"], "ippp", d.get("ippp",{}).get("value"))
PY
done; done | tee $O/summary.txt
