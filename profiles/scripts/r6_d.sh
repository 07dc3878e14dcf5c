#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_frame.py::test_multi_reference_p_pictures tests/test_gpu_configs.py::test_anchor_with_three_past_anchors_encoder_tools tests/test_gpu_configs.py::test_rdoq_in_the_pixel_path -x -q 2>&1 | tail -4 | tee $O/d_pytest.txt
for i in 1 2; do timeout 300 python bench.py --leg hot --streams 1 --hier-b 8 --steps 48 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hot hier8 1 stream:', d['value'], 'pictures/s')"; done | tee $O/d_bench.txt
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default:', d['value'], 'fps', d['psnr_y'], 'dB; ippp', d.get('ippp',{}).get('value'))" | tee -a $O/d_bench.txt
