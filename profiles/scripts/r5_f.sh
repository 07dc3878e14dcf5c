#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/f; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rc.py tests/test_gpu_configs.py -q -m gpu -k "config5" 2>&1 | tail -12 > $O/pytest_f.txt
cat $O/pytest_f.txt
