cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final6; mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_stream.py tests/test_gpu_enc_api.py -q -k "enc_ or decoder_verified_fixture" 2>&1 | tail -6 > $O/pytest_fixture.txt; tail -2 $O/pytest_fixture.txt
python - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
for W,H,seed,abc,pan in ((1920,1080,42,(37,53,19),(5,3)),(3840,2160,7,(67,91,33),(8,5))):
    base=make_clip(W,H,17,seed=seed,abc=abc,pan=pan)
    order=list(range(17))+list(range(15,0,-1))
    with open(f'/dev/shm/clip_{W}.yuv','wb') as f:
        for t in range(128): f.write(base[order[t%len(order)]].tobytes())
PY
{
echo "# ks265enc only, final state of round 3 (+ vector propagation, stage A2); reference rows: r02_same_clips.txt (same clips, same box type)"
for cfg in "1920 1080 slow 27" "3840 2160 slow 27"; do set -- $cfg
 for extra in "" "-bframes 0"; do
  echo "## $1x$2 -preset $3 -rc 0 -qp $4 -iper 128 $extra"
  for dq in -2 0 2 4; do q=$(( $4 + dq ))
   echo "ks265enc -qp $q: $(./ks265codec_amd/ks265enc -i /dev/shm/clip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset $3 -rc 0 -qp $q -iper 128 $extra -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')"
  done
 done
done
} > $O/same_clips.txt 2>&1
rm -f /dev/shm/clip_*.yuv /dev/shm/o.265
cut -c1-40,150-215 $O/same_clips.txt
timeout 100 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json; echo
for n in 0 1 2; do timeout 40 python bench.py --leg hot --streams 1 --steps 20 --warmup 5 --propagate $n --no-cpu-baseline > $O/hot_prop$n.json 2>/dev/null; python - <<PY
import json
try:
    d=json.load(open('$O/hot_prop$n.json')); print('rounds $n:', d.get('value'), 'fps', d['roofline']['stages_ms'])
except Exception as e: print('rounds $n: no line', e)
PY
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
