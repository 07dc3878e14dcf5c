#!/bin/bash
# round 6, call A: -ref0 on the device - the new / changed GPU tests, the same-clip table, a bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_configs.py::test_anchor_with_three_past_anchors_encoder_tools tests/test_gpu_rc.py tests/test_gpu_enc_api.py tests/test_gpu_frame.py::test_multi_reference_p_pictures -x -q 2>&1 | tail -15 > $O/a_pytest.txt; cat $O/a_pytest.txt
bash tools/r6_same_clips.sh same_clips_ref0
timeout 600 python bench.py --no-cpu-baseline 2>$O/a_bench.err | grep '^{' | tail -1 > $O/a_bench.json; head -c 1500 $O/a_bench.json; echo; tail -3 $O/a_bench.err
