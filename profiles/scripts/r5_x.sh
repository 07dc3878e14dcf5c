#!/bin/bash
# round 5, run x: the round's last state once more: GPU suite, smoke, the driver's bench command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/x; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 400 python bench.py 2>$O/bench.err | grep '^{' | tail -1 > $O/bench_line_default_steps128.json
python - <<PY
import json
d=json.load(open("$O/bench_line_default_steps128.json"))
print("value", d["value"], d["psnr_y"], d["steps"], d["config"].get("gop_lanes"), "ippp", d["ippp"]["value"], "cpu", d["cpu_baseline"]["value"], "roofline", d["roofline"]["frac"], d["roofline"]["counters_stale"])
PY
cat $O/pytest_gpu.txt; tail -1 $O/smoke.txt
