#!/bin/bash
# round-4 measurement artefacts in ONE gpurun call: everything on the kernel set of THIS snapshot, every profiles/*.json stamped with the kernel
# source hash (ks265codec_amd/build.py source_sha) and the git head handed in through KS265_GIT_HEAD; results land in gpurun_out/r04/, the builder copies them to profiles/.
# usage (builder container):  KS265_GIT_HEAD=$(git rev-parse --short HEAD) gpurun --timeout 900 -- "KS265_GIT_HEAD=$KS265_GIT_HEAD bash tools/r4_profiles.sh"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "head ${KS265_GIT_HEAD:-?} kernel_src_sha $(python -c 'from ks265codec_amd.build import source_sha; print(source_sha())')" > $O/stamp.txt
cd /tmp
# ---- 1. kernel traces: the hot path alone on one stream (the kernels' own durations), and the default command (the driver's)
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hot1 -o kt -- python $R/bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt_hot1/kt_results.db > $O/kernel_stats_hot_1stream.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_def -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt_def/kt_results.db > $O/kernel_stats_default_whole_run.txt
# ---- 2. HBM traffic: separate PMC passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE never together with other counters)
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
rm -f $O/hbm_traffic.json
python $R/tools/hbm_traffic.py $(ls $O/pmc_f/*counter_collection.csv | head -1) $(ls $O/pmc_w/*counter_collection.csv | head -1) 3840x2160 $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1
# ---- 3. SQ counters: instruction mix (VALU issue fraction) and - north_star - the LDS bank-conflict counters
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
rm -f $O/sq_counters.json
python $R/tools/sq_issue.py $(ls $O/pmc_sq/*counter_collection.csv | head -1) 3840x2160 $O/sq_counters.json > $O/sq_issue.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/sq_summary.py $(ls $O/pmc_sq2/*counter_collection.csv | head -1) > $O/sq_counters.txt 2>&1
rm -rf $O/pmc_f $O/pmc_w $O/pmc_sq $O/pmc_sq2 $O/kt_hot1 $O/kt_def
# ---- 4. the bench lines on the stamped counters (the driver's command last, with the fresh profiles/ files in place)
cp $O/hbm_traffic.json $O/sq_counters.json $R/profiles/ 2>/dev/null
cd $R
timeout 150 python bench.py --leg hot --streams 1 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_hot_1stream.json
timeout 200 python bench.py --hier-b 8 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_hier8.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_default.err | grep '^{' | tail -1 > $O/bench_line_default.json
# ---- 5. the lookahead operators (csrc/lookahead_ops.hip) timed, and the whole GPU suite on this snapshot
cd /tmp; timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt_la -o kt -- python $R/tools/la_ops_bench.py > $O/la_ops.txt 2>&1
python $R/tools/rocpd_stats.py $O/kt_la/kt_results.db 2>/dev/null | grep -E 'kernel|aq_|cutree_' >> $O/la_ops.txt; rm -rf $O/kt_la
cd $R; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/pytest_gpu.txt
ls -la $O; head -c 600 $O/bench_line_default.json; echo; head -22 $O/kernel_stats_hot_1stream.txt; cat $O/hbm_traffic.txt; head -14 $O/sq_counters.txt | cut -c1-200
# ---- 6. what the lookahead at the input costs (round 4: not waited for; uploads + analysis on one stream created first): the bench's hierarchical-B leg without it, with the
#         default (slice types, grid pictures only) and with -lookahead 8 (every picture analysed); IPPP with -lookahead 8; and the CLI on a 2160p clip whose pan puts anchors
#         8 apart at the edge of the search window (257 pictures): bytes and PSNR without and with the default
cd $R
{ for la in 0 -1 8; do echo -n "hierarchical-B 8, --lookahead $la: "; timeout 200 python bench.py --leg encoded --no-cpu-baseline --steps 24 --hier-b 8 --lookahead $la 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'pictures/s,', round(d['config']['bytes_per_picture']), 'B/picture,', d['psnr_y'], 'dB')"; done
  for la in 0 8; do echo -n "IPPP, --lookahead $la: "; timeout 200 python bench.py --leg encoded --no-cpu-baseline --steps 24 --lookahead $la 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'pictures/s')"; done; } > $O/lookahead_cost.txt 2>&1
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from ks265codec_amd.synth import make_clip
b = make_clip(3840, 2160, 17, seed=7, abc=(67, 91, 33), pan=(8, 5))
o = list(range(17)) + list(range(15, 0, -1))
with open('/dev/shm/mg.yuv', 'wb') as f:
    for t in range(257): f.write(b[o[t % len(o)]].tobytes())
PY
for extra in "-lookahead 0" ""; do echo "== ks265enc 3840x2160 257 pictures -preset slow -qp 27 -iper 128 (default GOP) $extra"; ./ks265codec_amd/ks265enc -i /dev/shm/mg.yuv -wdt 3840 -hgt 2160 -fr 50 -preset slow -rc 0 -qp 27 -iper 128 $extra -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'lookahead|bitrate, psnr'; done >> $O/lookahead_cost.txt 2>&1
rm -f /dev/shm/mg.yuv /dev/shm/o.265
cat $O/lookahead_cost.txt
