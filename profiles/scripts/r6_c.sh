#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_stream.py tests/test_gpu_configs.py::test_config3_2160p_encoder_tools tests/test_gpu_configs.py::test_anchor_with_three_past_anchors_encoder_tools -x -q 2>&1 | tail -6 | tee $O/c_pytest.txt
timeout 300 python bench.py --leg hot --streams 1 --hier-b 8 --steps 48 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hot hier8 1 stream:', d['value'], 'pictures/s; key picture', d['config']['key_picture_ms'])" | tee $O/c_bench.txt
cp ks265codec_amd/libks265hip.so /tmp/keep.so
KS_VARIANT=scratch/variants/libks265hip_clk.so timeout 600 python scratch/intra_clock.py 2>&1 | tail -60 | tee $O/intra_clock.txt
cp /tmp/keep.so ks265codec_amd/libks265hip.so
