#!/bin/bash
# round 6: where the uploads' stream is created among a lane's streams (hardware queues are shared in creation order) and what the default GOP / IPPP code then.  usage: gpurun -- 'bash tools/r6_upl.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for v in ${R6_UPL_VARIANTS:-"KS265_UPL_ORDER=4" "KS265_INPUT_COPY=1" "KS265_UPL_ORDER=0"}; do
  env $v timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/upl.json
  python - "$v" <<PY
import json, sys
d = json.load(open("$O/upl.json"))
c, i = d["config"].get("caller_ms_per_picture", {}), d.get("ippp", {}).get("caller_ms_per_picture", {})
print(sys.argv[1], "| hier", d["value"], "input", c.get("input_copy"), "enq", c.get("enqueue"), "out", c.get("output"), "| ippp", d.get("ippp", {}).get("value"), "input", i.get("input_copy"))
PY
done 2>&1 | tee $O/upl_order.txt
