"""TEST INFRASTRUCTURE (builder container only) — drive the reference binary's own `_c`
kernels through the LD_PRELOAD shim (probe_shim.c) and collect their outputs.

Never imported by the product, by `-m gpu` tests, by smoke() or by bench.py: it needs
/root/reference, which does not exist on the GPU box.  Its only products are the
fixture files under tests/golden/ (see gen_golden.py).

Addresses are virtual addresses in /root/reference/ubuntu_x64/appencoder (non-PIE,
based at 0x400000 — SURVEY.md §0, Appendix B).
"""
from __future__ import annotations

import os
import shutil
import struct
import subprocess
import tempfile

import numpy as np

REF_DIR = "/root/reference/ubuntu_x64"
HERE = os.path.dirname(os.path.abspath(__file__))

# reference kernel addresses (nm -C -n appencoder)
ADDR = {
    "sad_c": 0x47AE30, "sad4_c": 0x47AE90, "sad3_c": 0x47B060, "sad4blk_8x8_c": 0x4CEE30,
    "sse_c4": 0x47B230, "sse_c8": 0x47B270, "sse_c16": 0x47B2C0, "sse_c32": 0x47B310, "sse_c64": 0x47B360,
    "had_c": 0x47B680,
    "dst4": 0x4C2250, "dct4": 0x4C2210, "dct8": 0x4C2290, "dct16": 0x4C22D0, "dct32": 0x4C2310,
    "quant4": 0x4A9DE0, "quant8": 0x4A9E70, "quant16": 0x4A9EA0, "quant32": 0x4A9ED0,
    "quant_block": 0x4A9CF0, "get_base_quant_param": 0x4A9C90,
    "dequant_block": 0x439210, "get_base_dequant_param": 0x4391D0,
    "idst4": 0x448C40, "idct4": 0x448F60, "idct8": 0x449200, "idct16": 0x449880, "idct32": 0x44A0D0,
    "idct8_opt": 0x44B0B0, "idct16_opt": 0x44BD20, "idct32_opt": 0x44DC20,
    "idct4_dc": 0x434190, "idst4_dc": 0x434240, "idct8_dc": 0x4359A0, "idct16_dc": 0x4358C0, "idct32_dc": 0x4357E0,
    "edge_luma_ver": 0x403630, "edge_luma_hor": 0x4038C0,
    "chroma_ver": 0x403C50, "chroma_hor": 0x403D10,
    "luma_hor_8to8": 0x40E4F0, "luma_hor_8to16": 0x40EB80, "luma_ver_8to8": 0x40F0C0, "luma_ver_8to16": 0x40F950,
    "luma_ver_16to8": 0x4100B0, "luma_ver_16to16": 0x4109B0,
    "chroma_hor_8to8": 0x4111C0, "chroma_hor_8to16": 0x411310, "chroma_ver_8to8": 0x411430,
    "chroma_ver_8to16": 0x4115A0, "chroma_ver_16to8": 0x4116E0, "chroma_ver_16to16": 0x411850,
    "sao_bo": 0x43E4E0, "sao_bo_uv": 0x43E380, "sao_eo0": 0x43E650, "sao_eo1": 0x43E970,
    "sao_eo2": 0x43EDC0, "sao_eo3": 0x43EF70,
    "stat_bo_eo01": 0x4AE9C0, "stat_bo_eo01_luma": 0x4AEB20, "stat_bo_eo01_chroma": 0x4AEB50,
    "stat_eo01": 0x4AEBE0,
}


class Buf:
    """A caller-owned buffer handed to the reference kernel (copied in, copied back)."""

    def __init__(self, arr: np.ndarray):
        self.arr = np.ascontiguousarray(arr).copy()
        self.out: np.ndarray | None = None

    def at(self, byte_off: int) -> tuple["Buf", int]:
        return (self, int(byte_off))


class Call:
    def __init__(self, addr: int, args: list):
        self.addr = addr
        self.args = args
        self.ret: int | None = None
        self.bufs: list[Buf] = []
        for a in args:
            b = a[0] if isinstance(a, tuple) else a
            if isinstance(b, Buf) and not any(b is x for x in self.bufs):
                self.bufs.append(b)


class RefProbe:
    """Batch calls, run the reference binary once, fill in results."""

    def __init__(self):
        if not os.path.exists(os.path.join(REF_DIR, "appencoder")):
            raise RuntimeError("reference binary not available (expected in builder container only)")
        self.tmp = tempfile.mkdtemp(prefix="ks265probe_")
        shutil.copy(os.path.join(REF_DIR, "appencoder"), self.tmp)
        os.chmod(os.path.join(self.tmp, "appencoder"), 0o755)
        subprocess.check_call(["gcc", "-O1", "-w", "-shared", "-fPIC", "-o", os.path.join(self.tmp, "probe_shim.so"),
                               os.path.join(HERE, "probe_shim.c")])
        self.calls: list[Call] = []

    def call(self, name_or_addr, *args) -> Call:
        addr = ADDR[name_or_addr] if isinstance(name_or_addr, str) else name_or_addr
        c = Call(addr, list(args))
        self.calls.append(c)
        return c

    def run(self) -> None:
        job = os.path.join(self.tmp, "job.bin")
        out = os.path.join(self.tmp, "out.bin")
        with open(job, "wb") as f:
            f.write(struct.pack("<II", 0x4250534B, len(self.calls)))
            for c in self.calls:
                f.write(struct.pack("<QII", c.addr, len(c.args), len(c.bufs)))
                for b in c.bufs:
                    raw = b.arr.tobytes()
                    f.write(struct.pack("<I", len(raw)))
                    f.write(raw)
                for a in c.args:
                    if isinstance(a, tuple) or isinstance(a, Buf):
                        b, off = a if isinstance(a, tuple) else (a, 0)
                        idx = [i for i, x in enumerate(c.bufs) if x is b][0]
                        f.write(struct.pack("<IIq", 1, idx, off))
                    else:
                        f.write(struct.pack("<IIq", 0, 0, int(a)))
        env = dict(os.environ, LD_PRELOAD=os.path.join(self.tmp, "probe_shim.so"), KS265_PROBE_JOB=job, KS265_PROBE_OUT=out)
        r = subprocess.run([os.path.join(self.tmp, "appencoder"), "-v"], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"probe failed rc={r.returncode}: {r.stderr[-400:]}")
        if r.stderr.strip():
            raise RuntimeError("probe reported: " + r.stderr[-400:])
        data = open(out, "rb").read()
        pos = 0
        for c in self.calls:
            (c.ret,) = struct.unpack_from("<q", data, pos)
            pos += 8
            for b in c.bufs:
                n = b.arr.nbytes
                b.out = np.frombuffer(data[pos:pos + n], dtype=b.arr.dtype).reshape(b.arr.shape).copy()
                pos += n
        assert pos == len(data)
        self.calls = []

    def close(self):
        shutil.rmtree(self.tmp, ignore_errors=True)
