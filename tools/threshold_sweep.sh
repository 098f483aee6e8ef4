# bench.py under a few settings of the join's routing thresholds (segments with fewer probing
# queries / members go to the thin-segment filter): HS_JOIN_MIN_Q, HS_JOIN_MIN_M.  Run on the GPU box.
for cfg in "3 16" "2 16" "1 16" "2 4" "1 4" "1 1" "2 16" "3 16" "1 16" "2 4"; do
  set -- $cfg
  HS_JOIN_MIN_Q=$1 HS_JOIN_MIN_M=$2 python bench.py --no-cpu-baseline --recall-queries 0 > gpurun_out/thr_$1_$2.json 2> gpurun_out/thr.err
  python -c "
import json,sys; b=json.load(open('gpurun_out/thr_$1_$2.json')); r=b['roofline']; print('min_q=$1 min_m=$2', round(b['ms_per_step'],3), round(b['phases_ms_per_step']['verify'],3), r['work_items_per_step'], r['pairs_streamed_per_step'])"
done
