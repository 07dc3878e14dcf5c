#!/bin/bash
# round 6, call E: lanes, -rdoq 1 / -sao 3 in bytes and pictures/s, the straight clips
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for l in 2 3 4; do KS265_GOP_LANES=$l timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $l:', d['value'], 'fps', d['psnr_y'], 'dB')"; done | tee $O/e_lanes.txt
python - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
make_clip(1920,1080,65,seed=42,abc=(37,53,19),pan=(5,3)).tofile('/dev/shm/c.yuv')
PY
{ for opt in "" "-rdoq 1" "-sao 3" "-bframes 0" "-bframes 0 -rdoq 1" "-bframes 0 -sao 3"; do
 echo "1920x1080 65 pictures -preset slow -qp 27 $opt: $(./ks265codec_amd/ks265enc -i /dev/shm/c.yuv -wdt 1920 -hgt 1080 -fr 50 -preset slow -rc 0 -qp 27 -iper 128 $opt -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'Total Frames.*FPS|bitrate, psnr' | tr '\n' ' ')"
done; } | tee $O/e_rdoq_sao.txt
rm -f /dev/shm/c.yuv /dev/shm/o.265
bash tools/r6_straight_clips.sh straight_clips
