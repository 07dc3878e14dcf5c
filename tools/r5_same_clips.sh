R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/y; mkdir -p $O; cd $R
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
echo "# ks265enc only, END of round 5 (round 4's rate-distortion path + the joint refinement after the CU decision; two GOP lanes by default for the default GOP); reference rows: r02_same_clips.txt (same clips, same box type)"
for cfg in "1920 1080 slow 27" "3840 2160 slow 27"; do set -- $cfg
 for extra in "" "-bframes 0"; do
  echo "## $1x$2 -preset $3 -rc 0 -qp $4 -iper 128 $extra"
  for dq in -2 0 2 4; do q=$(( $4 + dq ))
   echo "ks265enc -qp $q: $(./ks265codec_amd/ks265enc -i /dev/shm/clip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset $3 -rc 0 -qp $q -iper 128 $extra -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')"
  done
 done
done
} > $O/same_clips_final.txt 2>&1
rm -f /dev/shm/clip_*.yuv /dev/shm/o.265
cat $O/same_clips_final.txt | cut -c1-200
