#!/bin/bash
# round 5, run e: several reference pictures per list in B pictures (parity on the device, the host, the CLI at config 5's command line)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/e; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -q -m gpu 2>&1 | tail -12 > $O/pytest_e.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "config5 or hier" 2>&1 | tail -12 >> $O/pytest_e.txt
timeout 900 python -m pytest tests/test_gpu_rc.py -q -m gpu 2>&1 | tail -12 >> $O/pytest_e.txt
cat $O/pytest_e.txt
