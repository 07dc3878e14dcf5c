#!/bin/bash
# round 6: what the cuTree pass costs at 2160p (config 4's command line), and where its time goes.  usage: gpurun --timeout 900 -- 'bash tools/r6_crf.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06/crf; mkdir -p $O; cd $R
export TMPDIR=/tmp
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from ks265codec_amd.synth import make_clip
c = make_clip(3840, 2160, 33, seed=7, abc=(67, 91, 33), pan=(8, 5))
order = list(range(33)) + list(range(31, 0, -1))
with open('/tmp/c2160.yuv', 'wb') as f:
    for t in range(129):
        f.write(c[order[t % len(order)]].tobytes())
PY
E=$R/ks265codec_amd/ks265enc
for ct in 1 0; do
  for rep in 1 2; do
    $E -i /tmp/c2160.yuv -wdt 3840 -hgt 2160 -fr 50 -preset slow -rc 3 -crf 24 -bframes 3 -iper 128 -cutree $ct -b /tmp/o_$ct.265 -log 1 2>&1 | grep -E "Total Frames|bitrate, psnr|cuTree" | tr '\n' ' ' >> $O/fps.txt; echo " [cutree $ct]" >> $O/fps.txt
  done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $E -i /tmp/c2160.yuv -wdt 3840 -hgt 2160 -fr 50 -preset slow -rc 3 -crf 24 -bframes 3 -iper 128 -b /tmp/o.265 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt/kt_results.db > $O/kernel_stats_crf24_b3.txt 2>&1
rm -rf $O/kt
cat $O/fps.txt; grep -E "cfc_|cutree|pad_plane|downsample|qoff|aq_|KERNEL|name" $O/kernel_stats_crf24_b3.txt | head -20; head -12 $O/kernel_stats_crf24_b3.txt | cut -c1-160
