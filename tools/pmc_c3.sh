#!/bin/bash
# PMC passes over the configs[2]-shape bench (run on the GPU box): tools/pmc_join.sh with that workload.
# usage: tools/pmc_c3.sh <tag> "<counter group 1>" "<counter group 2>" ...
export HS_BENCH_ARGS="--db-size 100000000 --L 32 --K 20 --W 160 --queries 125000"
exec bash $GRAFT_REPO_ROOT/tools/pmc_join.sh "$@"
