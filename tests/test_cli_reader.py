"""CPU: the CLI's read-ahead thread (ks265codec_amd/host/ks265_cli.c) against a stub of the encoder API (tests/cli_stub_encoder.c) - the stub checks that every
picture arrives complete, in order and with its pts; no GPU involved."""
from __future__ import annotations

import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def cli(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    exe = str(d / "cli_stub")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "ks265codec_amd", "host", "ks265_cli.c"),
                           os.path.join(HERE, "cli_stub_encoder.c"), "-o", exe, "-lpthread"])
    return exe


def _clip(path, W, H, n, tail=b""):
    with open(path, "wb") as f:
        for i in range(n):
            a = np.full(W * H * 3 // 2, i & 255, np.uint8)
            a[:4] = np.frombuffer(np.uint32(i).tobytes(), np.uint8)
            f.write(a.tobytes())
        f.write(tail)


@pytest.mark.parametrize("n,frms,expect", [(37, None, 37), (37, 10, 10), (37, 0, 0), (1, None, 1), (0, None, 0), (5, 9, 5)])
def test_reader_thread_delivers_every_picture_in_order(cli, tmp_path, n, frms, expect):
    W, H = 64, 48
    path = tmp_path / "in.yuv"
    _clip(path, W, H, n, tail=b"xx")                      # a partial trailing picture is ignored, as before
    args = [cli, "-i", str(path), "-wdt", str(W), "-hgt", str(H)] + (["-frms", str(frms)] if frms is not None else [])
    for _ in range(5):                                    # a few runs each: the hand-over is between two threads
        r = subprocess.run(args, capture_output=True, text=True, timeout=30)
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"stub saw {expect} pictures" in r.stdout and f"Total Frames: {expect}," in r.stdout and "H265 encoder passed!!!" in r.stdout, r.stdout
