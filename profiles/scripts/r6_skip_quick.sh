#!/bin/bash
# quick look after a kernel change: the touched GPU tests, kernel trace of the pyramid hot leg, the default bench line (with and without the skip pass)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/skipq; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_enc_api.py tests/test_gpu_stream.py -q -x -k "skip or cli_stream_equals" 2>&1 | tail -4 > $O/pytest.txt; cat $O/pytest.txt
one() {
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hier -o kt -- python $R/bench.py --leg hot --streams 1 --hier-b 8 --steps 48 --no-cpu-baseline > $O/bench_line_hot_hier8_1stream_$1.json 2>/dev/null
python $R/tools/rocpd_stats.py $O/kt_hier/kt_results.db > $O/kernel_stats_hier8_hot_1stream_$1.txt
rm -rf $O/kt_hier
cd $R
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_default.err | grep '^{' | tail -1 > $O/bench_line_default_$1.json
grep -E "skip_pass|timed region" $O/kernel_stats_hier8_hot_1stream_$1.txt | cut -c1-160; python -c "
import json,sys; d=json.load(open('$O/bench_line_default_$1.json')); print('$1 value', d['value'], 'ippp', d['ippp']['value'], 'psnr', d['psnr_y'], d['ippp'].get('bytes_per_picture'))"
}
one cur
KS265_SKIP_RD=0 one noskip
