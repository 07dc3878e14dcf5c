R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_frame.py tests/test_gpu_stream.py tests/test_gpu_enc_api.py tests/test_gpu_configs.py -m gpu -x -q -k "b_pictures or stream or cli or fuzz or replayed or decoder" 2>&1 | tail -4 | cut -c1-300
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.log 2>&1; grep '^{' $O/bench_driver_like.log | tail -1 > $O/bench_line_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02f/bench_line_default.json').read()); c=d['config']
print('value', d['value'], 'psnr', d['psnr_y'], 'kbps', c['kbps_at_50fps'], c['windows'], 'hot', d['hot_path']['value'], 'cpu', d['cpu_baseline']['value'])
PY
