#!/bin/bash
# round 5, run d: -part 1 in B pictures on the device (parity), neighbourhood work predictor (timeline), me_subpel with strided CTU groups (experiment build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/d; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_stream.py -q -m gpu -k "config5 or part or hier" 2>&1 | tail -12 > $O/pytest_part.txt
timeout 600 python -m pytest tests/test_gpu_rc.py -q -m gpu -k "config5" 2>&1 | tail -8 >> $O/pytest_part.txt
timeout 300 python tools/me_trace.py build/variants/libks265hip_trace.so > $O/me_trace.txt 2>&1
timeout 200 python tools/subpel_time.py > $O/subpel.txt 2>&1
timeout 200 python tools/subpel_time.py build/variants/libks265hip_strided.so >> $O/subpel.txt 2>&1
cat $O/pytest_part.txt; grep -E "picture|sum of WG" $O/me_trace.txt | cut -c1-330; grep -v amdgpu $O/subpel.txt
