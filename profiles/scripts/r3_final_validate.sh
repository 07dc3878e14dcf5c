cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final5; mkdir -p $O
# 1. the new stage alone; on failure dump its records and stop (keeps GPU budget for one more attempt)
timeout 400 python -m pytest tests/test_gpu_frame.py -q -x -k "vector_propagation" 2>&1 | tail -30 > $O/prop_test.txt; tail -4 $O/prop_test.txt
if grep -q "failed\|error" $O/prop_test.txt || ! grep -q " passed" $O/prop_test.txt; then
  timeout 200 python tools/prop_diag.py $O > $O/prop_diag.txt 2>&1; tail -30 $O/prop_diag.txt; exit 1
fi
# 2. the whole GPU suite
timeout 1200 python -m pytest tests -m gpu -q --maxfail=4 2>&1 | tail -40 > $O/pytest.txt; tail -5 $O/pytest.txt
if grep -q "failed" $O/pytest.txt; then exit 2; fi
# 3. bench line + smoke
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json; echo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# 4. cost of the stage: the device-resident path with one shard, 0 / 1 / 2 rounds
for n in 0 1 2; do KS265_PROPAGATE=$n timeout 200 python bench.py --leg hot --streams 1 --steps 20 --warmup 5 --propagate $n > $O/hot_prop$n.json 2>/dev/null; python - <<PY
import json
try:
    d=json.load(open('$O/hot_prop$n.json')); print('rounds $n:', d.get('value'), 'fps', d['roofline']['stages_ms'])
except Exception as e: print('rounds $n: no line', e)
PY
done
# 5. same clips
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
cut -c1-210 $O/same_clips.txt
