cd $GRAFT_REPO_ROOT; O=gpurun_out/r3mg; mkdir -p $O
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
b=make_clip(3840,2160,9,seed=7,abc=(67,91,33),pan=(8,5))
o=list(range(9))+list(range(7,0,-1))
with open('/dev/shm/mg.yuv','wb') as f:
    for t in range(65): f.write(b[o[t%len(o)]].tobytes())
PY
for extra in "" "-lookahead 8"; do for q in 27 29; do
 echo "2160p 65 pictures default GOP $extra qp $q: $(./ks265codec_amd/ks265enc -i /dev/shm/mg.yuv -wdt 3840 -hgt 2160 -fr 50 -preset slow -rc 0 -qp $q -iper 128 $extra -threads 32 -psnr 1 -b /dev/shm/o.265 | grep -E 'lookahead:|bitrate, psnr|pure' | tr '\n' ' ')"
done; done > $O/minigop_2160.txt 2>&1
cat $O/minigop_2160.txt; rm -f /dev/shm/mg.yuv /dev/shm/o.265
