#!/bin/bash
# round 5, run p: two GOP lanes by default for the pyramid GOPs: the encoder-level GPU tests and the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/p; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_enc_api.py tests/test_gpu_rc.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -8 > $O/pytest_p.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_default.err | grep '^{' | tail -1 > $O/bench_line_default.json
python - <<PY
import json
d=json.load(open("$O/bench_line_default.json"))
print("value", d["value"], d["psnr_y"], d["config"].get("gop_lanes"), d["config"]["windows"], "ippp", d["ippp"]["value"], "cpu", d["cpu_baseline"]["value"], "hot", d["hot_path"]["value"])
PY
cat $O/pytest_p.txt; tail -5 $O/bench_default.err | cut -c1-300
