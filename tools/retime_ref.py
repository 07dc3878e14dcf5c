"""re-time the reference CLI encoder on >= 128-frame clips of the SURVEY §8(d) generator (builder container, 8 vCPU)."""
import json, os, re, subprocess, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from ks265codec_amd.synth import make_clip
T = '/tmp/ref'
CFG = [("1280x720 veryfast qp32", 1280, 720, 43, (37, 53, 19), (5, 3), ["-preset", "veryfast", "-rc", "0", "-qp", "32", "-iper", "128"], 3),
       ("1920x1080 slow qp27", 1920, 1080, 42, (37, 53, 19), (5, 3), ["-preset", "slow", "-rc", "0", "-qp", "27", "-iper", "128"], 2),
       ("3840x2160 slow qp27", 3840, 2160, 7, (67, 91, 33), (8, 5), ["-preset", "slow", "-rc", "0", "-qp", "27", "-iper", "128"], 1)]
N = 128
res = []
for name, W, H, seed, abc, pan, args, runs in CFG:
    yuv = f"{T}/clip_{W}.yuv"
    if not os.path.exists(yuv):
        base = make_clip(W, H, 17, seed=seed, abc=abc, pan=pan)        # 17 distinct pictures, ping-pong to 128 (continuous motion)
        order = list(range(17)) + list(range(15, 0, -1))
        with open(yuv, "wb") as f:
            for t in range(N):
                f.write(base[order[t % len(order)]].tobytes())
    for th in (1, 8):
        fps_l = []
        for r in range(runs if th == 8 or W < 3840 else 1):
            out = subprocess.run([f"{T}/appencoder", "-i", yuv, "-wdt", str(W), "-hgt", str(H), "-fr", "50", *args, "-threads", str(th), "-psnr", "1", "-b", f"{T}/o.265"],
                                 capture_output=True, text=True, cwd=T).stdout
            m = re.search(r"FPS:\s*([0-9.]+)", out); b = re.search(r"bitrate, psnr:\s*([0-9.]+)\s+([0-9.]+)", out)
            fps_l.append(float(m.group(1)))
        rec = dict(config=name, frames=N, threads=th, fps_runs=fps_l, fps_median=float(np.median(fps_l)), kbps=float(b.group(1)), psnr_y=float(b.group(2)))
        print(rec, flush=True); res.append(rec)
json.dump(res, open('/root/repo/gpurun_out/ref_retime.json', 'w'), indent=1)
