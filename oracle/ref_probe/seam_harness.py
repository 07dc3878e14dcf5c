#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (builder container only): the seam harness of SURVEY.md §7.1 / §A.6.

Runs the reference CLI encoder twice on the same synthetic clip — untouched, and with its operator tables patched to the CPU
oracle's kernels (seam_shim.c) — and requires the two .265 files to be byte-identical.  Writes tests/golden/seam_report.json.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/ubuntu_x64/appencoder"

CONFIGS = [
    dict(name="416x240 veryfast qp32", w=416, h=240, frames=8, seed=1234, abc=(17, 23, 9), args=["-preset", "veryfast", "-rc", "0", "-qp", "32", "-iper", "128"]),
    dict(name="416x240 slow qp27", w=416, h=240, frames=6, seed=1234, abc=(17, 23, 9), args=["-preset", "slow", "-rc", "0", "-qp", "27", "-iper", "128"]),
    dict(name="416x240 slow qp27 bframes 3", w=416, h=240, frames=9, seed=99, abc=(17, 23, 9), args=["-preset", "slow", "-rc", "0", "-qp", "27", "-iper", "128", "-bframes", "3"]),
    # rate control on: adaptive quantisation calls acEnergyPlane (config 4 of BASELINE.json in miniature)
    dict(name="416x240 slow crf24 bframes 3", w=416, h=240, frames=9, seed=7, abc=(17, 23, 9), args=["-preset", "slow", "-rc", "3", "-crf", "24", "-iper", "128", "-bframes", "3"]),
]


def run(only: str | None = None) -> dict:
    from ks265codec_amd.synth import make_clip
    if not os.path.exists(REF):
        raise RuntimeError("reference binary not available")
    tmp = tempfile.mkdtemp(prefix="ks265seam_")
    try:
        shutil.copy(REF, tmp)
        enc = os.path.join(tmp, "appencoder")
        os.chmod(enc, 0o755)
        shim = os.path.join(tmp, "seam_shim.so")
        subprocess.check_call(["gcc", "-O2", "-w", "-shared", "-fPIC", "-o", shim, os.path.join(HERE, "seam_shim.c"),
                               os.path.join(ROOT, "oracle", "ks265_oracle.c"), os.path.join(ROOT, "oracle", "ks265_intra_oracle.c"), "-ldl", "-lpthread"])
        report = {"reference": "ubuntu_x64/appencoder (libqycodec V2.6.1.3), -threads 1", "runs": []}
        for cfg in CONFIGS:
            clip = make_clip(cfg["w"], cfg["h"], cfg["frames"], seed=cfg["seed"], abc=cfg["abc"])
            yuv = os.path.join(tmp, "in.yuv")
            clip.tofile(yuv)
            outs = {}
            counts = {}
            for mode in ("plain", "seam"):
                out = os.path.join(tmp, mode + ".265")
                env = dict(os.environ)
                if mode == "seam":
                    env.update(LD_PRELOAD=shim, KS265_SEAM="1", KS265_SEAM_COUNTS=os.path.join(tmp, "counts.txt"))
                    if only:
                        env["KS265_SEAM_ONLY"] = only
                cmd = [enc, "-i", yuv, "-wdt", str(cfg["w"]), "-hgt", str(cfg["h"]), "-fr", "50", *cfg["args"], "-threads", "1", "-b", out]
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=tmp)
                if r.returncode != 0 or not os.path.exists(out):
                    raise RuntimeError(f"encoder failed ({mode}): {r.stdout[-300:]} {r.stderr[-300:]}")
                outs[mode] = open(out, "rb").read()
                if mode == "seam":
                    counts = {k: int(v) for k, v in (l.split() for l in open(os.path.join(tmp, "counts.txt")))}
            report["runs"].append({"config": cfg["name"], "args": " ".join(cfg["args"]), "frames": cfg["frames"],
                                   "bytes": len(outs["plain"]), "md5_plain": hashlib.md5(outs["plain"]).hexdigest(),
                                   "md5_seam": hashlib.md5(outs["seam"]).hexdigest(), "identical": outs["plain"] == outs["seam"],
                                   "oracle_calls": counts})
        return report
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    rep = run(sys.argv[1] if len(sys.argv) > 1 else None)
    for r in rep["runs"]:
        print(r["config"], "identical" if r["identical"] else "DIFFERENT", r["bytes"], "bytes;", {k: v for k, v in r["oracle_calls"].items() if v})
    if len(sys.argv) <= 1:
        json.dump(rep, open(os.path.join(ROOT, "tests", "golden", "seam_report.json"), "w"), indent=1)
