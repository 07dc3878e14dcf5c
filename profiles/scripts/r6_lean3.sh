#!/bin/bash
# lean B pictures: the non-reference pictures alone (KS265_LEAN_B=3) against + the reference B pictures with near references (the default, 1): parity tests, headline, straight-clip table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/lean; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_rc.py tests/test_gpu_enc_api.py -q -m gpu -x -k "tools_per_picture or crf or config5 or adaptive_q or pyramid or decoder_reproduces" 2>&1 | tail -4
for v in 3 1; do
  KS265_LEAN_B=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_default_lean$v.json
  python - <<PY
import json; d=json.load(open("$O/bench_line_default_lean$v.json")); print("lean $v value", d["value"], "psnr", d["psnr_y"], "kbps", d["config"]["kbps_at_50fps"], "ippp", d["ippp"]["value"])
PY
  KS265_LEAN_B=$v bash tools/r6_straight_clips.sh straight_lean$v > /dev/null 2>&1; cat $R/gpurun_out/r06/straight_lean${v}_equal_psnr.txt
done
