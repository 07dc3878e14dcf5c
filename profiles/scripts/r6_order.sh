#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/order; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_frame.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_default.json
python - <<PY
import json; d=json.load(open("$O/bench_line_default.json")); print("value", d["value"], "psnr", d["psnr_y"], "kbps", d["config"]["kbps_at_50fps"], "ippp", d["ippp"]["value"], "hot", d["hot_path"]["value"])
PY
( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --leg hot --streams 1 --hier-b 8 --steps 48 --no-cpu-baseline > $O/bench_line_hot_hier8.json 2>/dev/null )
python tools/rocpd_stats.py $O/kt/kt_results.db > $O/kernel_stats_hier8.txt; rm -rf $O/kt; head -4 $O/kernel_stats_hier8.txt | cut -c1-140; grep -E "me_order|me_score" $O/kernel_stats_hier8.txt | cut -c1-140; grep -o '"value": [0-9.]*' $O/bench_line_hot_hier8.json | head -1
