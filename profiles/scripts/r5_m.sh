#!/bin/bash
# round 5, run m: SQ counters of a hierarchical-B picture's kernels (hot leg, one stream)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/m; mkdir -p $O; cd /tmp
export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- python $R/bench.py --leg hot --hier-b 8 --steps 16 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/sq_summary.py $(ls $O/pmc_sq2/*counter_collection.csv | head -1) > $O/sq_counters_hier8.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --leg hot --hier-b 8 --steps 16 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/sq_summary.py $(ls $O/pmc_sq/*counter_collection.csv | head -1) > $O/sq_insts_hier8.txt 2>&1
rm -rf $O/pmc_sq $O/pmc_sq2
cut -c1-220 $O/sq_counters_hier8.txt | head -14; cut -c1-220 $O/sq_insts_hier8.txt | head -14
