#!/bin/bash
# round 5, run b: (1) per-work-group timeline of me_int_kernel (experiment build), (2) B pictures with list 1's search on a side stream against KS265_B_SERIAL=1: parity tests,
# hot-leg kernel traces and rates, the encoded leg
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/b; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python tools/me_trace.py build/variants/libks265hip_trace.so > $O/me_trace.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_frame.py tests/test_gpu_stream.py -q -m gpu -x 2>&1 | tail -6 > $O/pytest_b.txt
for mode in par ser; do
  if [ $mode = ser ]; then export KS265_B_SERIAL=1; else unset KS265_B_SERIAL; fi
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_$mode -o kt -- python $R/bench.py --leg hot --streams 1 --hier-b 8 --steps 48 --no-cpu-baseline > $O/hot_hier8_$mode.json 2>/dev/null
  python $R/tools/rocpd_stats.py $O/kt_$mode/kt_results.db > $O/kernel_stats_hier8_hot_1stream_$mode.txt; rm -rf $O/kt_$mode
  cd $R
  timeout 200 python bench.py --leg encoded --no-cpu-baseline --hier-b 8 --steps 24 2>/dev/null | grep '^{' | tail -1 > $O/enc_hier8_$mode.json
done
unset KS265_B_SERIAL
cat $O/me_trace.txt; cat $O/pytest_b.txt
for mode in par ser; do echo "== $mode"; python -c "
import json,sys
for fn in ('$O/hot_hier8_$mode.json','$O/enc_hier8_$mode.json'):
    for l in open(fn):
        if l.startswith('{'):
            d=json.loads(l); print(fn.split('/')[-1], d['value'], 'fps', d['psnr_y'], 'dB')"; head -12 $O/kernel_stats_hier8_hot_1stream_$mode.txt | cut -c1-150; done
