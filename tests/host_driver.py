"""Drives the SDK-compatible encoder API of a libks265enc build given by $KS265_STUB_LIB (tests/test_host_pipeline_cpu.py: the host linked against the CPU stand-in of
the device library, tests/hip_stub.c) as fast as the API takes pictures, and prints one JSON line: stream md5, lanes, pts of the coded pictures in output order ...
argv: repo root, pictures, key period, bframes, width, height [, output file]"""
import ctypes as C, hashlib, json, os, sys
ROOT, N, iper, bframes, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
out_path = sys.argv[7] if len(sys.argv) > 7 else None
import numpy as np
LAY = json.load(open(os.path.join(ROOT, "tests", "golden", "qy265_layout.json")))
lib = C.CDLL(os.environ["KS265_STUB_LIB"]); lib.QY265EncoderOpen.restype = C.c_void_p
class YUV(C.Structure): _fields_ = [("iWidth", C.c_int), ("iHeight", C.c_int), ("pData", C.POINTER(C.c_ubyte) * 3), ("iStride", C.c_int * 3)]
class Picture(C.Structure): _fields_ = [("iSliceType", C.c_int), ("poc", C.c_int), ("pts", C.c_longlong), ("dts", C.c_longlong), ("yuv", C.POINTER(YUV))]
class Nal(C.Structure): _fields_ = [("naltype", C.c_int), ("tid", C.c_int), ("iSize", C.c_int), ("pts", C.c_longlong), ("pPayload", C.POINTER(C.c_ubyte))]
rng = np.random.default_rng(3)
clip = rng.integers(0, 256, (11, W * H * 3 // 2), dtype=np.uint8)
cfg = (C.c_uint8 * LAY["sizeof_config"])()
assert lib.QY265ConfigDefaultPreset(cfg, b"medium", None, os.environ.get("KS_TEST_LATENCY", "default").encode()) == 0
if os.environ.get("KS_TEST_SCENECUT"):
    assert lib.ks265_enc_set_default(b"scenecut", int(os.environ["KS_TEST_SCENECUT"])) == 0
for k, v in (("wdt", W), ("hgt", H), ("fr", 50), ("rc", int(os.environ.get("KS_TEST_RC", "0"))), ("br", int(os.environ.get("KS_TEST_BR", "1000"))), ("qp", 34), ("iper", iper), ("bframes", bframes), ("threads", 5), ("psnr", 1), ("log", 3), ("lookahead", int(os.environ.get("KS_TEST_LOOKAHEAD", "-1"))), ("aq", int(os.environ.get("KS_TEST_AQ", "0"))), ("ref", int(os.environ.get("KS_TEST_REF", "1")))) + ((("ref0", int(os.environ["KS_TEST_REF0"])),) if os.environ.get("KS_TEST_REF0") else ()) + ((("rdoq", int(os.environ["KS_TEST_RDOQ"])),) if os.environ.get("KS_TEST_RDOQ") else ()):
    assert lib.QY265ConfigParse(cfg, k.encode(), str(v).encode()) == 0
err = C.c_int(0)
h = C.c_void_p(lib.QY265EncoderOpen(cfg, C.byref(err))); assert h.value, hex(err.value & 0xFFFFFFFF)
nal, nn, pic, outp, yuv = C.POINTER(Nal)(), C.c_int(0), Picture(), Picture(), YUV()
hdr_entries = None
if os.environ.get("KS_TEST_HEADERS"):                     # QY265EncoderEncodeHeaders: VPS, SPS, PPS as three entries
    assert lib.QY265EncoderEncodeHeaders(h, C.byref(nal), C.byref(nn)) == 0
    hdr_entries = [(nal[i].naltype, nal[i].iSize, bytes(C.string_at(nal[i].pPayload, min(6, nal[i].iSize))).hex()) for i in range(nn.value)]
yuv.iWidth, yuv.iHeight = W, H
yuv.iStride[0], yuv.iStride[1], yuv.iStride[2] = W, W // 2, W // 2
pic.yuv = C.pointer(yuv)
md, pts, types, bs = hashlib.md5(), [], [], bytearray()
errors, err_at = 0, []                                     # KS_TEST_CONTINUE_ON_ERROR: calls that reported a failure (the caller goes on feeding pictures)
maxdelay = 0
zc = 0
def take(check=True):
    for i in range(nn.value):
        if nal[i].iSize <= 0: continue                      # a failed picture's empty entry
        b = C.string_at(nal[i].pPayload, nal[i].iSize); md.update(b); bs.extend(b); types.append(nal[i].naltype)
        if nal[i].naltype < 32: pts.append(nal[i].pts)
    if check and nn.value and pts:
        assert outp.poc == pts[-1], ("the output picture's display index", outp.poc, pts[-1])    # pts = display index in this driver
strided = bool(os.environ.get("KS_TEST_STRIDE"))
if strided:                                                # planes with padded rows: the library copies row by row
    pad = 24
    yuv.iStride[0], yuv.iStride[1], yuv.iStride[2] = W + pad, W // 2 + pad, W // 2 + pad
    planes = [np.zeros((H, W + pad), np.uint8), np.zeros((H // 2, W // 2 + pad), np.uint8), np.zeros((H // 2, W // 2 + pad), np.uint8)]
cuts = [int(x) for x in os.environ.get("KS_TEST_CUTS", "").split(",") if x]
if cuts:                                                   # scenes: one base picture per scene + a little noise per picture, a new base at every cut
    scene, base = -1, None
    frames = []
    for t in range(N):
        if t == 0 or t in cuts:
            scene += 1; base = clip[scene % 11].astype(np.int16)
        frames.append(np.clip(base + rng.integers(-3, 4, base.shape), 0, 255).astype(np.uint8))
ramp = [int(x) for x in os.environ.get("KS_TEST_RAMP", "").split(":") if x]          # start:end:step - one scene whose brightness rises by `step` per picture inside [start, end)
if ramp:
    base = rng.integers(48, 112, W * H * 3 // 2).astype(np.int16)
    frames = [np.clip(base + ramp[2] * (min(max(t, ramp[0]), ramp[1]) - ramp[0]) + rng.integers(-3, 4, base.shape), 0, 255).astype(np.uint8) for t in range(N)]
one = np.zeros(W * H * 3 // 2, np.uint8) if os.environ.get("KS_TEST_ONE_BUFFER") else None    # the SDK's own callers (encoderwrapper.c:367-379) refill ONE buffer for every picture
for t in range(N):
    fr = frames[t] if (cuts or ramp) else clip[t % 11]
    if one is not None:
        one[:] = fr; fr = one
    if strided:
        planes[0][:, :W] = fr[:W * H].reshape(H, W); planes[0][:, W:] = t & 255
        planes[1][:, :W // 2] = fr[W * H:W * H * 5 // 4].reshape(H // 2, W // 2); planes[2][:, :W // 2] = fr[W * H * 5 // 4:].reshape(H // 2, W // 2)
        for k in range(3): yuv.pData[k] = C.cast(planes[k].ctypes.data, C.POINTER(C.c_ubyte))
    elif os.environ.get("KS_TEST_ZEROCOPY") and lib.ks265_enc_acquire_input(h, C.byref(yuv)) == 0:      # the picture is produced straight into one of the encoder's own buffers
        C.memmove(yuv.pData[0], fr.ctypes.data, W * H * 3 // 2); zc += 1
    else:
        yuv.iStride[0], yuv.iStride[1], yuv.iStride[2] = W, W // 2, W // 2
        for k, off in enumerate((0, W * H, W * H * 5 // 4)): yuv.pData[k] = C.cast(fr.ctypes.data + off, C.POINTER(C.c_ubyte))
    pic.pts = t
    if os.environ.get("KS_TEST_RECONFIG") and t in (40, 90):
        assert lib.QY265ConfigParse(cfg, b"qp", str(30 + t // 40).encode()) == 0
        lib.QY265EncoderReconfig(h, cfg)
    rc = lib.QY265EncoderEncodeFrame(h, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(outp), 0)
    if rc and os.environ.get("KS_TEST_EXPECT_ERROR"):       # a device failure: the API reports it, closing the encoder must not hang
        lib.QY265EncoderClose(h)
        print(json.dumps({"error": rc & 0xFFFFFFFF, "at": t}))
        sys.exit(0)
    if rc and os.environ.get("KS_TEST_CONTINUE_ON_ERROR"):
        errors += 1; err_at.append(t); take(False); continue
    assert rc == 0, hex(rc & 0xFFFFFFFF)
    if one is not None: one[:] = 0xA5                       # the call has returned: the buffer is the caller's again (scribbled over before it is refilled)
    take(errors == 0)
    maxdelay = max(maxdelay, lib.QY265EncoderDelayedFrames(h))
    if os.environ.get("KS_TEST_KEYREQ") and t in (17, 18, 40): lib.QY265EncoderKeyFrameRequest(h)
if os.environ.get("KS_TEST_CLOSE_EARLY"):                 # no flush: pictures are still in flight on every thread when the handle is closed
    lib.QY265EncoderClose(h)
    print(json.dumps({"closed_early": True, "vcl": len(pts)}))
    sys.exit(0)
calls = 0
while lib.QY265EncoderDelayedFrames(h):
    rc = lib.QY265EncoderEncodeFrame(h, C.byref(nal), C.byref(nn), None, C.byref(outp), 0)
    if rc and os.environ.get("KS_TEST_EXPECT_ERROR"):       # GOP lanes buffer deeply: a short clip may be fed completely before the failed picture is through
        lib.QY265EncoderClose(h)
        print(json.dumps({"error": rc & 0xFFFFFFFF, "at": N}))
        sys.exit(0)
    if rc and os.environ.get("KS_TEST_CONTINUE_ON_ERROR"):
        errors += 1; err_at.append(N + calls); take(False); calls += 1; continue
    assert rc == 0, hex(rc & 0xFFFFFFFF)
    take(errors == 0); calls += 1
    assert calls < 100000
lanes = lib.ks265_enc_lanes(h)
lib.QY265EncoderClose(h)
if out_path: open(out_path, "wb").write(bytes(bs))
print(json.dumps({"hdr": hdr_entries, "md5": md.hexdigest(), "lanes": lanes, "pts": pts, "idr": types.count(19), "vcl": len(pts), "bytes": len(bs), "maxdelay": maxdelay, "flush_calls": calls, "zero_copy": zc, "errors": errors, "err_at": err_at}))
