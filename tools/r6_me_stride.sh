#!/bin/bash
# round 6: the row pitch of me_int_kernel's LDS window against its bank conflicts.  Variants are built in the builder container into scratch/variants/
# (KS_WIN_STRIDE=N); here each one replaces the library for one run of the device-resident IPPP leg: kernel duration (HIP events) + SQ LDS counters.  usage: gpurun -- 'bash tools/r6_me_stride.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06/me_stride; mkdir -p $O; cd $R
export TMPDIR=/tmp
cp ks265codec_amd/libks265hip.so /tmp/base.so
for v in base ws212 ws220 ws228; do
  if [ $v = base ]; then cp /tmp/base.so ks265codec_amd/libks265hip.so; else cp scratch/variants/libks265hip_$v.so ks265codec_amd/libks265hip.so; fi
  timeout 150 python bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 > $O/line_$v.json
  (cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/pmc_$v -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/sq_summary.py $(ls $O/pmc_$v/*counter_collection.csv | head -1) 2>/dev/null | grep -E "^kernel|me_int_kernel|me_subpel" > $O/sq_$v.txt
  rm -rf $O/pmc_$v
  python - $v <<PY
import json, sys
d = json.load(open("$O/line_%s.json" % sys.argv[1]))
r = d.get("roofline", {})
print(sys.argv[1], "me_int avg ms", r.get("avg_launch_ms"), "stages", r.get("stages_ms"), "fps", d.get("value"))
PY
  cat $O/sq_$v.txt | cut -c1-220
done 2>&1 | tee $O/summary.txt
cp /tmp/base.so ks265codec_amd/libks265hip.so
