#!/usr/bin/env python3
"""How many pictures per second ONE calling thread can push through ONE handle: the encoder host linked against the CPU stand-in of the device
library in its KS265_STUB_FAST mode (a "device" that costs the host next to nothing), 3840x2160 pictures, 1 / 2 / 4 / 8 "GPUs" behind the handle (KS265_GPUS = one GOP lane per
device).  With enough lanes the scheduler threads and writers keep up and the calling thread is what is left: its time per picture = input copy into pinned memory (12.4 MB,
shared with three helper threads) + enqueue + collecting output.  Prints one line per lane count and the thread budget of each.  No GPU needed; run it on the box whose host
matters (`gpurun -- python tools/caller_ceiling.py`).  usage: caller_ceiling.py [pictures per lane = 192] [bframes = 0] [zero]"""
import ctypes as C, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests")
DRIVER = r'''
import ctypes as C, json, os, sys, time
import numpy as np
ROOT, N, iper, bframes, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), 3840, 2160
LAY = json.load(open(os.path.join(ROOT, "tests", "golden", "qy265_layout.json")))
lib = C.CDLL(os.environ["KS265_STUB_LIB"]); lib.QY265EncoderOpen.restype = C.c_void_p
class YUV(C.Structure): _fields_ = [("iWidth", C.c_int), ("iHeight", C.c_int), ("pData", C.POINTER(C.c_ubyte) * 3), ("iStride", C.c_int * 3)]
class Picture(C.Structure): _fields_ = [("iSliceType", C.c_int), ("poc", C.c_int), ("pts", C.c_longlong), ("dts", C.c_longlong), ("yuv", C.POINTER(YUV))]
class Nal(C.Structure): _fields_ = [("naltype", C.c_int), ("tid", C.c_int), ("iSize", C.c_int), ("pts", C.c_longlong), ("pPayload", C.POINTER(C.c_ubyte))]
class Stats(C.Structure): _fields_ = [("frames", C.c_long), ("bytes", C.c_longlong), ("sse", C.c_double * 3), ("gpu_ms", C.c_double), ("host_write_ms", C.c_double), ("in_copy_ms", C.c_double),
                                      ("submit_ms", C.c_double), ("output_ms", C.c_double), ("lat_gpu_ms", C.c_double), ("lat_queue_ms", C.c_double), ("key_wall_ms", C.c_double), ("key_cpu_ms", C.c_double),
                                      ("keys", C.c_long), ("occ", C.c_long * 4), ("submit_wait_ms", C.c_double)]
clip = np.random.default_rng(3).integers(0, 256, (5, W * H * 3 // 2), dtype=np.uint8)
cfg = (C.c_uint8 * LAY["sizeof_config"])()
assert lib.QY265ConfigDefaultPreset(cfg, b"slow", None, b"default") == 0
for k, v in (("wdt", W), ("hgt", H), ("fr", 50), ("rc", 0), ("qp", 27), ("iper", iper), ("bframes", bframes), ("threads", int(os.environ.get("KS_THREADS", "0"))), ("psnr", 0), ("log", 3)):
    assert lib.QY265ConfigParse(cfg, k.encode(), str(v).encode()) == 0
err = C.c_int(0)
h = C.c_void_p(lib.QY265EncoderOpen(cfg, C.byref(err))); assert h.value, hex(err.value & 0xFFFFFFFF)
nal, nn, pic, outp, yuv = C.POINTER(Nal)(), C.c_int(0), Picture(), Picture(), YUV()
yuv.iWidth, yuv.iHeight = W, H
yuv.iStride[0], yuv.iStride[1], yuv.iStride[2] = W, W // 2, W // 2
pic.yuv = C.pointer(yuv)
ZERO = bool(os.environ.get("KS_ZERO_COPY"))
def feed(t):
    fr = clip[t % 5]
    if ZERO and lib.ks265_enc_acquire_input(h, C.byref(yuv)) == 0: pass      # zero-copy input: the application produces the picture in the encoder's buffer (its own work: not timed here)
    else:
        yuv.iStride[0], yuv.iStride[1], yuv.iStride[2] = W, W // 2, W // 2
        for k, off in enumerate((0, W * H, W * H * 5 // 4)): yuv.pData[k] = C.cast(fr.ctypes.data + off, C.POINTER(C.c_ubyte))
    pic.pts = t
    assert lib.QY265EncoderEncodeFrame(h, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(outp), 0) == 0
lanes = lib.ks265_enc_lanes(h)
warm = lanes * iper
for t in range(warm): feed(t)
s0 = Stats(); lib.ks265_enc_get_stats(h, C.byref(s0))
t0 = time.perf_counter()
for t in range(warm, warm + N): feed(t)
dt = time.perf_counter() - t0
s1 = Stats(); lib.ks265_enc_get_stats(h, C.byref(s1))
while lib.QY265EncoderDelayedFrames(h): assert lib.QY265EncoderEncodeFrame(h, C.byref(nal), C.byref(nn), None, C.byref(outp), 0) == 0
lib.QY265EncoderClose(h)
print(json.dumps({"lanes": lanes, "pictures": N, "seconds": dt, "fps_fed": N / dt, "input_copy_ms": (s1.in_copy_ms - s0.in_copy_ms) / N, "output_ms": (s1.output_ms - s0.output_ms) / N,
                  "call_ms": 1000 * dt / N}))
'''

def main():
    per_lane = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    bframes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tmp = tempfile.mkdtemp(prefix="ks265ceil_")
    so = os.path.join(tmp, "libks265enc_stub.so")
    host = os.path.join(ROOT, "ks265codec_amd", "host")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-I", os.path.join(ROOT, "include"), "-shared", "-o", so, os.path.join(host, "ks265_enc.c"), os.path.join(host, "ks265_stream.c"),
                           os.path.join(HERE, "hip_stub.c"), "-L", os.path.join(ROOT, "oracle"), "-lks265_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread", "-lm"])
    ncpu = os.cpu_count()
    print(f"# host: {ncpu} hardware threads; 3840x2160, -preset slow -rc 0 -iper 64 -bframes {bframes}, stand-in device in KS265_STUB_FAST mode, {per_lane} timed pictures per lane")
    zero = len(sys.argv) > 3 and sys.argv[3] == "zero"
    if zero: print("# zero-copy input (ks265_enc_acquire_input): the picture is produced in the encoder's pinned buffer, the call copies nothing")
    for g in (1, 2, 4, 8):
        env = dict(os.environ, KS265_STUB_LIB=so, KS265_STUB_FAST="1", KS265_GPUS=str(g), KS265_GOP_LANES="1", KS265_PINNED_MB="2048", **({"KS_ZERO_COPY": "1"} if zero else {}))
        r = subprocess.run([sys.executable, "-c", DRIVER, ROOT, str(per_lane * g), "64", str(bframes)], capture_output=True, text=True, timeout=900, env=env)
        if r.returncode:
            print(f"GPUs {g}: failed: {r.stderr[-300:]}"); continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        # threads of the handle: per lane min(32, (ncpu - 3 L) / L) slice writers + scheduler + dispatcher; three copy helpers; the caller
        th = max(2, min(32, (ncpu - 3 * d["lanes"]) // d["lanes"])) if d["lanes"] > 1 else min(32, ncpu)
        print(f"GPUs {g}: lanes {d['lanes']}  fed {d['fps_fed']:8.1f} pictures/s  per call {d['call_ms']:.3f} ms (input copy {d['input_copy_ms']:.3f}, collecting output {d['output_ms']:.3f})  "
              f"threads: {d['lanes']} x ({th} writers + 2) + 3 copy helpers + caller = {d['lanes'] * (th + 2) + 4}")

if __name__ == "__main__":
    main()
