#!/bin/bash
# lean B pictures (ks265_frame_set_picture_tools): parity tests, then the default bench line with and without (KS265_LEAN_B=0), same box, back to back
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/lean; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_rc.py tests/test_gpu_enc_api.py -q -m gpu -x -k "tools_per_picture or crf or config5 or adaptive_q or pyramid or lanes or api_call or decoder_reproduces or skip_pass" 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
for v in 1 0; do
  KS265_LEAN_B=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_default_lean$v.json
  python - <<PY
import json; d=json.load(open("$O/bench_line_default_lean$v.json")); print("lean $v value", d["value"], "psnr", d["psnr_y"], "kbps", d["config"]["kbps_at_50fps"], "ippp", d["ippp"]["value"], d["config"]["caller_ms_per_picture"]["input_copy"])
PY
done
