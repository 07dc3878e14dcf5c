#!/bin/bash
# configs[0] with this build's SAO rule (-sao 4) instead of the preset's level 3 (= the reference's decision restated): ks265enc rows only, against the reference rows of profiles/r06_config0_720p.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
make_clip(1280,720,128,seed=11,abc=(29,41,15),pan=(4,2)).tofile('/dev/shm/sclip_1280.yuv')
PY
{
for extra in "" "-bframes 0"; do
  echo "## 1280x720 -preset veryfast -rc 0 -qp 32 -iper 128 $extra"
  for dq in -2 0 2 4; do q=$(( 32 + dq ))
   echo "ks265enc -qp $q: $(./ks265codec_amd/ks265enc -i /dev/shm/sclip_1280.yuv -wdt 1280 -hgt 720 -fr 50 -preset veryfast -rc 0 -qp $q -iper 128 $extra $KS_EXTRA -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'Total|bitrate, psnr' | tr '\n' ' ')"
  done
done
} > $O/config0_extra.txt 2>&1
rm -f /dev/shm/sclip_1280.yuv /dev/shm/o.265
python tools/equal_psnr.py profiles/r06_config0_720p.txt $O/config0_extra.txt
