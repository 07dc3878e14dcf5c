#!/bin/bash
# round 5, run o: GOP lanes with the default GOP (1 / 2 / 3 lanes), and the device-resident hierarchical leg with 1 / 2 / 3 shards in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/o; mkdir -p $O; cd $R
export TMPDIR=/tmp
for L in 1 2 3; do
  KS265_GOP_LANES=$L timeout 300 python bench.py --leg encoded --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_lanes$L.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/bench_lanes$L.json").read().strip().splitlines()[-1])
print("lanes $L hier", d["value"], d["config"].get("gop_lanes"), "ippp", d.get("ippp",{}).get("value"))
PY
done | tee $O/summary.txt
for S in 1 2 3; do
  timeout 200 python bench.py --leg hot --hier-b 8 --streams $S --steps 48 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hot hier8 streams $S', d['value'])"
done | tee -a $O/summary.txt
