#!/bin/bash
# bits per picture and layer of the default GOP, the reference's -psnr 2 lines beside ks265enc's: 1080p and 2160p straight clips, qp 27.
# Lines: poc <tab> type <tab> bytes <tab> psnr-y ... <tab> qp (both encoders print the reference's format).  Layers by picture type and QP
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python - <<'PY'
import sys; sys.path.insert(0,'.')
from ks265codec_amd.synth import make_clip
for W,H,seed,abc,pan in ((1920,1080,42,(37,53,19),(5,3)),(3840,2160,7,(67,91,33),(8,5))):
    make_clip(W,H,128,seed=seed,abc=abc,pan=pan).tofile(f'/dev/shm/sclip_{W}.yuv')
PY
mkdir -p /tmp/ks_s && cp oracle/_ref/appencoder /tmp/ks_s/appencoder_s && chmod +x /tmp/ks_s/appencoder_s
for cfg in "1920 1080" "3840 2160"; do set -- $cfg
  ( cd /tmp/ks_s && ./appencoder_s -i /dev/shm/sclip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset slow -rc 0 -qp 27 -iper 128 -threads 64 -psnr 2 -b /dev/shm/r.265 > /tmp/ks_s/ref_$1.txt 2>&1 )
  ./ks265codec_amd/ks265enc -i /dev/shm/sclip_$1.yuv -wdt $1 -hgt $2 -fr 50 -preset slow -rc 0 -qp 27 -iper 128 -threads 32 -psnr 2 -b /dev/shm/o.265 > /tmp/ks_s/ours_$1.txt 2>&1
  python - $1 $2 <<'PY'
import re, sys
W = sys.argv[1]
def rows(path):
    out = []
    for ln in open(path, errors="ignore"):
        m = re.match(r"^(\d+)\t([IPB])\t(\d+)\t([\d.]+)\t[\d.]+\t[\d.]+\t(\d+)\s*$", ln)
        if m: out.append((int(m.group(1)), m.group(2), int(m.group(3)), float(m.group(4)), int(m.group(5))))
    return out
def layer(poc, t, q):
    return "key" if t == "I" else f"{t} qp {q}"                 # (the slice-type decision moves pictures between positions: picture type and QP name the layer - anchors Q + 1, B pictures + 2 / + 4)
print(f"## {W}x{sys.argv[2]} -preset slow -rc 0 -qp 27 -iper 128 (default GOP), 128 distinct pictures: bits per picture (mean), PSNR-Y (mean), pictures, share of the stream")
for name, path in (("reference", f"/tmp/ks_s/ref_{W}.txt"), ("ks265enc", f"/tmp/ks_s/ours_{W}.txt")):
    agg = {}
    for poc, t, b, p, q in rows(path):
        a = agg.setdefault(layer(poc, t, q), [0, 0, 0.0, set()]); a[0] += 1; a[1] += b; a[2] += p; a[3].add(q)
    tot = sum(a[1] for a in agg.values())
    print(f"{name:10s} " + "  ".join(f"{k}: {a[1] / a[0]:8.0f} bits {a[2] / a[0]:.2f} dB x{a[0]} ({100.0 * a[1] / tot:.0f} %)" for k, a in sorted(agg.items())) + f"   total {tot} bits")
PY
done > $O/layer_bytes.txt 2>&1
rm -rf /dev/shm/sclip_*.yuv /dev/shm/o.265 /dev/shm/r.265 /tmp/ks_s
cat $O/layer_bytes.txt | cut -c1-330
