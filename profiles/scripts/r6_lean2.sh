#!/bin/bash
# lean B pictures level 1 (tools) against level 2 (+ interMeHex instead of interMeUMH): headline and the device-resident pyramid's kernel trace, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/lean; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "tools_per_picture" 2>&1 | tail -3
for v in 1 2; do
  KS265_LEAN_B=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_default_lean$v.json
  python - <<PY
import json; d=json.load(open("$O/bench_line_default_lean$v.json")); print("lean $v value", d["value"], "psnr", d["psnr_y"], "kbps", d["config"]["kbps_at_50fps"], "ippp", d["ippp"]["value"])
PY
  ( cd /tmp; KS265_LEAN_B=$v timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hier$v -o kt -- python $R/bench.py --leg hot --streams 1 --hier-b 8 --steps 48 --no-cpu-baseline > $O/bench_line_hot_hier8_lean$v.json 2>/dev/null )
  python tools/rocpd_stats.py $O/kt_hier$v/kt_results.db > $O/kernel_stats_hier8_lean$v.txt; rm -rf $O/kt_hier$v
  head -8 $O/kernel_stats_hier8_lean$v.txt | cut -c1-140; grep -o '"value": [0-9.]*' $O/bench_line_hot_hier8_lean$v.json | head -1
done
