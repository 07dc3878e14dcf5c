#!/bin/bash
# round 5, run n: cfg.bi_refine = 2 (refinement for the chosen CUs): parity (stage by stage, fixtures' bytes, config mirrors), hot hier leg kernel times, default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/n; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frame.py -q -m gpu -k "b_pictures_match" 2>&1 | tail -6 > $O/pytest_n.txt
timeout 900 python -m pytest tests/test_gpu_stream.py -q -m gpu -k "bir2 or hier" 2>&1 | tail -6 >> $O/pytest_n.txt
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_rc.py tests/test_gpu_dqp.py tests/test_gpu_enc_api.py -q -m gpu 2>&1 | tail -8 >> $O/pytest_n.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hier -o kt -- python $R/bench.py --leg hot --streams 1 --hier-b 8 --steps 48 --no-cpu-baseline > $O/bench_line_hot_hier8_1stream.json 2>/dev/null
python $R/tools/rocpd_stats.py $O/kt_hier/kt_results.db > $O/kernel_stats_hier8_hot_1stream.txt; rm -rf $O/kt_hier
cd $R
timeout 300 python bench.py --leg encoded --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_enc.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$O/bench_enc.json").read().strip().splitlines()[-1])
print("hier", d["value"], d.get("psnr_y"), d["config"].get("kbps_at_50fps"), "ippp", d.get("ippp",{}).get("value"))
PY
cat $O/pytest_n.txt; head -12 $O/kernel_stats_hier8_hot_1stream.txt | cut -c1-150; grep "bi_" $O/kernel_stats_hier8_hot_1stream.txt | cut -c1-150
