#!/bin/bash
# same-clip table on clips WITHOUT repeats (round 6): 128 distinct pictures of SURVEY 8(d)'s generator - the ping-pong clips of profiles/r02_same_clips.txt repeat pictures 16 / 32 apart, which the
# reference's three-picture anchors (-ref0 3) and, since round 6, ours turn into near-free pictures; here both encoders see every picture once.  Reference (appencoder -threads 64) and
# ks265enc on the same box, default GOP and IPPP.  usage: gpurun -- bash tools/r6_straight_clips.sh [tag]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
TAG=${1:-straight_clips}
python - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
for W,H,seed,abc,pan in ((1920,1080,42,(37,53,19),(5,3)),(3840,2160,7,(67,91,33),(8,5))):
    make_clip(W,H,128,seed=seed,abc=abc,pan=pan).tofile(f'/dev/shm/sclip_{W}.yuv')
PY
{
echo "# 128 distinct pictures per clip (no ping-pong); head ${KS265_GIT_HEAD:-?}; $(nproc) host threads, one MI355X"
for cfg in "1920 1080 slow 27" "3840 2160 slow 27"; do set -- $cfg
 for extra in "" "-bframes 0"; do
  echo "## $1x$2 -preset $3 -rc 0 -qp $4 -iper 128 $extra"
  ( mkdir -p /tmp/ks_s && cd /tmp/ks_s && cp $R/oracle/_ref/appencoder ./appencoder_s && chmod +x ./appencoder_s && echo "reference appencoder -threads 64: $(./appencoder_s -i /dev/shm/sclip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset $3 -rc 0 -qp $4 -iper 128 $extra -threads 64 -psnr 1 -b /dev/shm/r.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')" )
  for dq in -2 0 2 4; do q=$(( $4 + dq ))
   echo "ks265enc -qp $q: $(./ks265codec_amd/ks265enc -i /dev/shm/sclip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset $3 -rc 0 -qp $q -iper 128 $extra -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')"
  done
 done
done
} > $O/$TAG.txt 2>&1
rm -rf /dev/shm/sclip_*.yuv /dev/shm/o.265 /dev/shm/r.265 /tmp/ks_s
cut -c1-200 $O/$TAG.txt
python tools/equal_psnr.py $O/$TAG.txt | tee $O/${TAG}_equal_psnr.txt
