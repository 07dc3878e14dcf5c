#!/bin/bash
# round 5, run w: the driver's N = 1 command once more, and the N = 2 launch path on the one GPU (both ranks on device 0, gloo for the barriers: functional check only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/w; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/b1.err | grep '^{' | tail -1 > $O/bench_n1.json
KS265_BENCH_BACKEND=gloo KS265_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/b2.err | grep '^{' | tail -1 > $O/bench_n2_one_device.json
python - <<PY
import json
for f in ("bench_n1","bench_n2_one_device"):
    try:
        d=json.load(open("$O/"+f+".json")); print(f, d["value"], d["n_gpus"], d["config"].get("gop_lanes"), d["ippp"]["value"], d["hot_path"]["value"])
    except Exception as e: print(f, "failed", e)
PY
tail -3 $O/b2.err | cut -c1-300
