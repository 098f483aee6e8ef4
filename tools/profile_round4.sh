#!/bin/bash
# Round-4 profile set (run on the GPU box from the repo root): kernel statistics of the default bench (C2) and of
# the secondary shape (configs[2], 10^6 queries in one batch and 125 k), PMC passes for the dominant kernel at C2
# (utilisation counters, then FETCH_SIZE and WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md
# prescribes) -> profiles/traffic_latest.json, FETCH / WRITE passes at the secondary shape ->
# profiles/traffic_secondary.json, then the bench line itself (which then carries both traffic figures).
# Usage: bash tools/profile_round4.sh r04_v2
TAG=$1
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
C3="--db-size 100000000 --L 32 --K 20 --W 160"
QUIET="--no-cpu-baseline --no-secondary --pcie-steps 0 --recall-queries 0 --planted-members 0 --general-steps 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $ROOT/bench.py --steps 20 --warmup 5 $QUIET > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_prof.err
cp $OUT/prof_$TAG/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats_c2.csv; rm -rf $OUT/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_c3 -- python3 $ROOT/bench.py $C3 --queries 1000000 --steps 8 --warmup 2 $QUIET > $OUT/${TAG}_bench_c3_1M_under_rocprof.json 2>> $OUT/${TAG}_prof.err
cp $OUT/prof_${TAG}_c3/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats_c3_1M.csv; rm -rf $OUT/prof_${TAG}_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_c3b -- python3 $ROOT/bench.py $C3 --queries 125000 --steps 8 --warmup 2 $QUIET > $OUT/${TAG}_bench_c3_125k_under_rocprof.json 2>> $OUT/${TAG}_prof.err
cp $OUT/prof_${TAG}_c3b/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats_c3_125k.csv; rm -rf $OUT/prof_${TAG}_c3b
cd $ROOT
bash tools/pmc_join.sh $TAG "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" > $OUT/${TAG}_pmc.log 2>&1
python3 tools/pmc_summarize.py $OUT/pmc_$TAG hs_join8x_kernel $OUT/${TAG}_pmc_join8x_kernel.json --traffic > $OUT/${TAG}_pmc_summary.txt
HS_BENCH_ARGS="$C3 --queries 1000000 --general-steps 0" bash tools/pmc_join.sh ${TAG}_sec "FETCH_SIZE" "WRITE_SIZE" >> $OUT/${TAG}_pmc.log 2>&1
python3 tools/pmc_secondary.py $OUT/pmc_${TAG}_sec 1000000 160 > $OUT/${TAG}_pmc_secondary.txt
cp profiles/traffic_latest.json $OUT/${TAG}_traffic_latest.json
cp profiles/traffic_secondary.json $OUT/${TAG}_traffic_secondary.json
rm -rf $OUT/pmc_$TAG/p*/ $OUT/pmc_${TAG}_sec/p*/
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_pmc_secondary.txt
