#!/bin/bash
# round-2 measurement artefacts in one call; everything lands in gpurun_out/r02/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
# ---- 1. same clips, same box, both encoders (the reference's CLI staged as oracle/_ref/appencoder)
python - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
for W,H,seed,abc,pan in ((1280,720,43,(37,53,19),(5,3)),(1920,1080,42,(37,53,19),(5,3)),(3840,2160,7,(67,91,33),(8,5))):
    base=make_clip(W,H,17,seed=seed,abc=abc,pan=pan)
    order=list(range(17))+list(range(15,0,-1))
    with open(f'/dev/shm/clip_{W}.yuv','wb') as f:
        for t in range(128): f.write(base[order[t%len(order)]].tobytes())
PY
{
echo "# 128-picture clips of SURVEY 8(d)'s generator (17 distinct pictures, ping-pong), both encoders on the same box: $(nproc) host threads ($(lscpu | grep 'Model name' | sed 's/.*: *//')), one MI355X"
for cfg in "1280 720 veryfast 32" "1920 1080 slow 27" "3840 2160 slow 27"; do set -- $cfg
 for extra in "" "-bframes 0"; do
  echo "## $1x$2 -preset $3 -rc 0 -qp $4 -iper 128 $extra"
  for th in 64; do
   echo "reference appencoder -threads $th: $(oracle/_ref/appencoder -i /dev/shm/clip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset $3 -rc 0 -qp $4 -iper 128 $extra -threads $th -psnr 1 -b /dev/shm/o_ref.265 2>&1 | grep -E 'FPS|bitrate, psnr' | tr '\n' ' ')"
  done
  echo "ks265enc -threads 32: $(./ks265codec_amd/ks265enc -i /dev/shm/clip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset $3 -rc 0 -qp $4 -iper 128 $extra -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')"
  for dq in 2 4 6; do q=$(( $4 + dq ))
   echo "ks265enc -qp $q: $(./ks265codec_amd/ks265enc -i /dev/shm/clip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset $3 -rc 0 -qp $q -iper 128 $extra -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'bitrate, psnr' | tr '\n' ' ')"
  done
 done
done
} > $O/same_clips.txt 2>&1
rm -f /dev/shm/clip_*.yuv /dev/shm/o.265 /dev/shm/o_ref.265
# ---- 2. bench lines
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.log 2>&1; grep '^{' $O/bench_driver_like.log | tail -1 > $O/bench_line_default.json
for i in 2 3; do timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --leg encoded 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_default_encoded_run$i.json; done
timeout 150 python bench.py --hier-b 8 --no-cpu-baseline > $O/bench_hier8.log 2>&1; grep '^{' $O/bench_hier8.log | tail -1 > $O/bench_line_hier8.json
timeout 150 python bench.py --leg hot --streams 1 --no-cpu-baseline > $O/bench_hot1.log 2>&1; grep '^{' $O/bench_hot1.log | tail -1 > $O/bench_line_hot_1stream.json
timeout 150 python bench.py --scaling strong --job-frames 1024 --out /dev/shm/job.265 > $O/bench_strong.log 2>&1; grep '^{' $O/bench_strong.log | tail -1 > $O/bench_line_strong_1gpu.json; rm -f /dev/shm/job.265
# ---- 3. kernel traces
cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $O/kt_hot1 -o kt -- python $R/bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt_hot1/kt_results.db > $O/kernel_stats_hot_1stream.txt
timeout 150 rocprofv3 --kernel-trace --stats -d $O/kt_def -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt_def/kt_results.db > $O/kernel_stats_default_whole_run.txt
# ---- 4. HBM traffic (separate PMC passes) and SQ counters
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
cp $R/profiles/hbm_traffic.json $O/hbm_traffic.json 2>/dev/null
python $R/tools/hbm_traffic.py $(ls $O/pmc_f/*counter_collection.csv | head -1) $(ls $O/pmc_w/*counter_collection.csv | head -1) 3840x2160 $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/sq_summary.py $(ls $O/pmc_sq/*counter_collection.csv | head -1) > $O/sq_counters.txt 2>&1
rm -rf $O/pmc_f $O/pmc_w $O/pmc_sq $O/kt_hot1 $O/kt_def
cd $R; timeout 200 python -m pytest tests/test_gpu_enc_api.py -x -q 2>&1 | tail -3 > $O/enc_api_tests.txt; cat $O/enc_api_tests.txt; ls -la $O
