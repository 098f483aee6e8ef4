#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel trace + stats of the default
# bench, PMC passes for the dominant kernel (utilisation counters, then FETCH_SIZE and WRITE_SIZE in
# passes of their own, as MI355X_MICROARCH.md prescribes), summaries copied to gpurun_out/<tag>_*.
# Usage: bash tools/profile_round.sh r02_v3
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --pcie-steps 0 --recall-queries 0 --planted-members 0 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_prof.err
cp $OUT/prof_$TAG/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats_c2.csv
# the same for the secondary block's shape (configs[2], one GPU's share): the two join kernels' durations there
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_c3 -- python3 $ROOT/bench.py --db-size 100000000 --L 32 --K 20 --W 160 --queries 125000 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --pcie-steps 0 --recall-queries 0 --planted-members 0 > $OUT/${TAG}_bench_c3_under_rocprof.json 2>> $OUT/${TAG}_prof.err
cp $OUT/prof_${TAG}_c3/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats_c3.csv
cd $ROOT
python3 bench.py > $OUT/${TAG}_bench_c2.json 2> $OUT/${TAG}_bench.err
bash tools/pmc_join.sh $TAG "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" > $OUT/${TAG}_pmc.log 2>&1
python3 tools/pmc_summarize.py $OUT/pmc_$TAG hs_join8x_kernel $OUT/${TAG}_pmc_join8x_kernel.json --traffic > $OUT/${TAG}_pmc_summary.txt
cp profiles/traffic_latest.json $OUT/${TAG}_traffic_latest.json
# the bench line again, now that profiles/traffic_latest.json carries this build's kernel hash
python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench_c2_with_traffic.json 2>> $OUT/${TAG}_bench.err
tail -5 $OUT/${TAG}_pmc_summary.txt
