#!/bin/bash
# what SAO is worth by the judged measure: ks265enc with the flags in $KS_EXTRA (e.g. -sao 0) on the straight clips, against the reference rows of profiles/r06_straight_clips.txt
# usage: gpurun -- 'KS_EXTRA="-sao 0" bash tools/r6_sao_off.sh tag'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
TAG=${1:-straight_sao0}
python - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
for W,H,seed,abc,pan in ((1920,1080,42,(37,53,19),(5,3)),(3840,2160,7,(67,91,33),(8,5))):
    make_clip(W,H,128,seed=seed,abc=abc,pan=pan).tofile(f'/dev/shm/sclip_{W}.yuv')
PY
{
echo "# ks265enc $KS_EXTRA; 128 distinct pictures per clip; head ${KS265_GIT_HEAD:-?}"
for cfg in "1920 1080 slow 27" "3840 2160 slow 27"; do set -- $cfg
 for extra in "" "-bframes 0"; do
  echo "## $1x$2 -preset $3 -rc 0 -qp $4 -iper 128 $extra"
  for dq in -2 0 2 4; do q=$(( $4 + dq ))
   echo "ks265enc -qp $q: $(./ks265codec_amd/ks265enc -i /dev/shm/sclip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset $3 -rc 0 -qp $q -iper 128 $extra $KS_EXTRA -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')"
  done
 done
done
} > $O/$TAG.txt 2>&1
rm -rf /dev/shm/sclip_*.yuv /dev/shm/o.265
python tools/equal_psnr.py profiles/r06_straight_clips.txt $O/$TAG.txt | tee $O/${TAG}_equal_psnr.txt
