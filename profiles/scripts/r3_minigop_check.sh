cd $GRAFT_REPO_ROOT; O=gpurun_out/r3mg; mkdir -p $O
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
make_clip(832,480,41,seed=7,abc=(37,53,19),pan=(8,5)).tofile('/dev/shm/mg.yuv')
PY
for extra in "" "-lookahead 8"; do
 echo "## ks265enc 832x480 41 pictures, default GOP $extra"
 ./ks265codec_amd/ks265enc -i /dev/shm/mg.yuv -wdt 832 -hgt 480 -fr 50 -preset slow -rc 0 -qp 27 -iper 128 $extra -threads 8 -psnr 2 -o /dev/shm/rec.yuv -b /dev/shm/o.265 > $O/log.txt 2>&1
 grep -E "lookahead:|bitrate, psnr|Total Frames" $O/log.txt | head -4
 awk '/^poc/{f=1;next} f&&NF>=7{printf "%s%s ", $1,$2} ' $O/log.txt | cut -c1-220; echo
 ./oracle/_ref/appdecoder -b /dev/shm/o.265 -o /dev/shm/dec.yuv -threads 4 > /dev/null 2>&1
 cmp /dev/shm/rec.yuv /dev/shm/dec.yuv && echo "DECODED == RECONSTRUCTION ($(stat -c %s /dev/shm/dec.yuv) bytes)"
done > $O/minigop.txt 2>&1
cat $O/minigop.txt
