#!/bin/bash
# round 5, run q: deeper lookahead queue for GOP lanes: default bench (two lanes for the pyramid), lanes 3, and KS265_LA_KEEP=5 for comparison
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/q; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --leg encoded --hier-b 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$tag.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
c=d["config"]; print("$tag", d["value"], c.get("gop_lanes"), c["windows"]["B"], c["caller_ms_per_picture"]["input_copy"], c["caller_ms_per_picture"]["output"])
PY
}
run default X=1
run keep5 KS265_LA_KEEP=5
run lanes3 KS265_GOP_LANES=3
run q8first GPU_MAX_HW_QUEUES=8
run q4 GPU_MAX_HW_QUEUES=4
timeout 600 python -m pytest tests/test_gpu_enc_api.py -q -m gpu 2>&1 | tail -3
