#!/bin/bash
# BASELINE.json configs[0] (1280x720 -preset veryfast -rc 0 -qp 32 -iper 128) as an equal-PSNR row: 128 distinct pictures of SURVEY 8(d)'s generator, the reference
# (appencoder -threads 1 as the config names it, and -threads 64) and ks265enc on the same box, default GOP and IPPP; KS265_SKIP_RD=2 (the skip pass on P pictures too) beside the default.
# usage: gpurun -- bash tools/r6_config0.sh [tag]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
TAG=${1:-config0_720p}
python - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
make_clip(1280,720,128,seed=11,abc=(29,41,15),pan=(4,2)).tofile('/dev/shm/sclip_1280.yuv')
PY
for v in 1 2; do
{
echo "# 128 distinct pictures (no ping-pong); head ${KS265_GIT_HEAD:-?}; KS265_SKIP_RD=$v; $(nproc) host threads, one MI355X"
for extra in "" "-bframes 0"; do
  echo "## 1280x720 -preset veryfast -rc 0 -qp 32 -iper 128 $extra"
  ( mkdir -p /tmp/ks_s && cd /tmp/ks_s && cp $R/oracle/_ref/appencoder ./appencoder_s && chmod +x ./appencoder_s
    [ $v = 1 ] && echo "# reference -threads 1: $(./appencoder_s -i /dev/shm/sclip_1280.yuv -wdt 1280 -hgt 720 -fr 50 -preset veryfast -rc 0 -qp 32 -iper 128 $extra -threads 1 -psnr 1 -b /dev/shm/r.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')"
    echo "reference appencoder -threads 64: $(./appencoder_s -i /dev/shm/sclip_1280.yuv -wdt 1280 -hgt 720 -fr 50 -preset veryfast -rc 0 -qp 32 -iper 128 $extra -threads 64 -psnr 1 -b /dev/shm/r.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')" )
  for dq in -2 0 2 4; do q=$(( 32 + dq ))
   echo "ks265enc -qp $q: $(KS265_SKIP_RD=$v ./ks265codec_amd/ks265enc -i /dev/shm/sclip_1280.yuv -wdt 1280 -hgt 720 -fr 50 -preset veryfast -rc 0 -qp $q -iper 128 $extra -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')"
  done
done
} > $O/${TAG}_skip$v.txt 2>&1
cut -c1-200 $O/${TAG}_skip$v.txt
python tools/equal_psnr.py $O/${TAG}_skip$v.txt | tee $O/${TAG}_skip${v}_equal_psnr.txt
done
rm -rf /dev/shm/sclip_*.yuv /dev/shm/o.265 /dev/shm/r.265 /tmp/ks_s
