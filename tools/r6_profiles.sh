#!/bin/bash
# round-6 measurement artefacts in ONE gpurun call, everything on the kernel set of THIS snapshot (profiles/*.json stamped with the kernel source hash + git head):
# kernel traces (hot leg IPPP one stream; hot leg hierarchical-B 8 one stream; the driver's default command), HBM traffic (FETCH / WRITE passes), SQ counters, bench lines, GPU suite.
# usage: KS265_GIT_HEAD=$(git rev-parse --short HEAD) gpurun --timeout 1500 -- "KS265_GIT_HEAD=$KS265_GIT_HEAD bash tools/r6_profiles.sh"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/prof; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "head ${KS265_GIT_HEAD:-?} kernel_src_sha $(python -c 'from ks265codec_amd.build import source_sha; print(source_sha())')" > $O/stamp.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hot1 -o kt -- python $R/bench.py --leg hot --streams 1 --steps 40 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt_hot1/kt_results.db > $O/kernel_stats_hot_1stream.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_hier -o kt -- python $R/bench.py --leg hot --streams 1 --hier-b 8 --steps 48 --no-cpu-baseline > $O/bench_line_hot_hier8_1stream.json 2>/dev/null
python $R/tools/rocpd_stats.py $O/kt_hier/kt_results.db > $O/kernel_stats_hier8_hot_1stream.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_def -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $O/kt_def/kt_results.db > $O/kernel_stats_default_whole_run.txt
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
rm -f $O/hbm_traffic.json
python $R/tools/hbm_traffic.py $(ls $O/pmc_f/*counter_collection.csv | head -1) $(ls $O/pmc_w/*counter_collection.csv | head -1) 3840x2160 $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
rm -f $O/sq_counters.json
python $R/tools/sq_issue.py $(ls $O/pmc_sq/*counter_collection.csv | head -1) 3840x2160 $O/sq_counters.json > $O/sq_issue.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- python $R/bench.py --leg hot --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/sq_summary.py $(ls $O/pmc_sq2/*counter_collection.csv | head -1) > $O/sq_counters.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq3 -o p -- python $R/bench.py --leg hot --hier-b 8 --steps 16 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/sq_summary.py $(ls $O/pmc_sq3/*counter_collection.csv | head -1) > $O/sq_counters_hier8.txt 2>&1
rm -rf $O/pmc_f $O/pmc_w $O/pmc_sq $O/pmc_sq2 $O/pmc_sq3 $O/kt_hot1 $O/kt_def $O/kt_hier
cp $O/hbm_traffic.json $O/sq_counters.json $R/profiles/ 2>/dev/null
cd $R
timeout 150 python bench.py --leg hot --streams 1 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_hot_1stream.json
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_default.err | grep '^{' | tail -1 > $O/bench_line_default.json
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
ls $O; head -c 700 $O/bench_line_default.json; echo; head -20 $O/kernel_stats_hot_1stream.txt | cut -c1-150; head -14 $O/kernel_stats_hier8_hot_1stream.txt | cut -c1-150; cat $O/hbm_traffic.txt | head -20; head -12 $O/sq_counters.txt | cut -c1-200; cat $O/pytest_gpu.txt
