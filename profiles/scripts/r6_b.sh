#!/bin/bash
# round 6, call B: -rdoq 1 and the reference's SAO decision on the device
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_configs.py::test_rdoq_in_the_pixel_path tests/test_gpu_configs.py::test_reference_sao_decision tests/test_gpu_rc.py::test_rdoq_command_line tests/test_gpu_rdoq.py -x -q 2>&1 | tail -25 > $O/b_pytest.txt; cat $O/b_pytest.txt
