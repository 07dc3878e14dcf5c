# the CLI run of tests/test_gpu_enc_api.py::test_cli_stream_equals_decoder_verified_fixture as a shell script (gpurun, a few seconds): ks265enc on the clip of the
# stream case enc_ippp_416x240_umh must write the stream of tests/golden/stream_md5.json (made on the CPU: oracle pipeline + writer, decoder-verified)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/r3fix
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from ks265codec_amd.synth import make_clip
name = "enc_ippp_416x240_umh"
make_clip(416, 240, 4, seed=len(name) * 7 + 416, abc=(17, 23, 9)).tofile('/dev/shm/fix.yuv')
PY
./ks265codec_amd/ks265enc -i /dev/shm/fix.yuv -wdt 416 -hgt 240 -fr 50 -preset slow -rc 0 -qp 27 -iper 128 -bframes 0 -threads 3 -psnr 2 -b /dev/shm/f.265 > gpurun_out/r3fix/log.txt 2>&1
python3 - <<'PY'
import hashlib, json
g = json.load(open('tests/golden/stream_md5.json'))["enc_ippp_416x240_umh"]
b = open('/dev/shm/f.265', 'rb').read()
print(f"expected {g['stream_md5']} {g['stream_bytes']} bytes; got {hashlib.md5(b).hexdigest()} {len(b)} bytes")
PY
grep -P "^\d+\t[IPB]" gpurun_out/r3fix/log.txt | cut -f1,2,7 | tr '\n' ' '; echo
