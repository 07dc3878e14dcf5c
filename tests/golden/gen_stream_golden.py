#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (builder container: needs the reference decoder).  For every case of tests/stream_cases.py: encode with the CPU oracle
pipeline, write the stream with the host writer, decode it with the reference's own decoder (ubuntu_x64/appdecoder) and require the decoded
pictures to equal the pipeline's reconstruction; then record MD5 of the stream and of every reconstructed picture in stream_md5.json.  The
fixture lets the GPU box (no decoder there) check that the HIP pipeline + writer produce the very streams that were decoder-verified here."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from stream_cases import CASES, make_stream, oracle_encoder  # noqa: E402

DEC = "/root/reference/ubuntu_x64/appdecoder"


def decode(bs: bytes, W: int, H: int, tmp: str) -> np.ndarray:
    dec = os.path.join(tmp, "appdecoder")
    if not os.path.exists(dec):
        shutil.copy(DEC, dec); os.chmod(dec, 0o755)
    open(os.path.join(tmp, "t.265"), "wb").write(bs)
    out = os.path.join(tmp, "t.yuv")
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([dec, "-b", "t.265", "-o", "t.yuv", "-threads", "1"], capture_output=True, text=True, cwd=tmp)
    if "decoder passed" not in r.stdout or not os.path.exists(out):
        raise RuntimeError("reference decoder failed: " + r.stdout[-300:] + r.stderr[-300:])
    return np.fromfile(out, np.uint8).reshape(-1, W * H * 3 // 2)


if __name__ == "__main__":
    tmp = tempfile.mkdtemp(prefix="ks265dec_")
    res = {}
    try:
        for name, c in CASES.items():
            bs, recs = make_stream(name, oracle_encoder(name))
            dec = decode(bs, c[0], c[1], tmp)
            assert len(dec) == len(recs), (name, len(dec), len(recs))
            for d in sorted(recs):
                assert (dec[d] == recs[d]).all(), f"{name}: decoded picture {d} differs from the pipeline's reconstruction"
            res[name] = {"stream_md5": hashlib.md5(bs).hexdigest(), "stream_bytes": len(bs), "decoder": "appdecoder V2.6.1.3: output == reconstruction",
                         "recon_md5": [hashlib.md5(recs[d].tobytes()).hexdigest() for d in sorted(recs)]}
            print(name, len(bs), "bytes,", len(recs), "pictures: decoded == reconstruction")
        json.dump(res, open(os.path.join(HERE, "stream_md5.json"), "w"), indent=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
