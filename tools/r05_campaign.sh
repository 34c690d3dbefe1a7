#!/bin/bash
# The measurement campaign behind profiles/r05_*: run a stage on the GPU box, everything lands under gpurun_out/r05/ with
# its final name (copy into profiles/ afterwards).   bash tools/r05_campaign.sh <stage>
set -e
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
case "$1" in
ee_tests)
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_configs.py -m gpu -x -q -k "early_exit or long_lists or walk_rows_many or random_db_vs_oracle or golden_db or random_config" --durations=8 > $O/ee_tests.txt 2>&1 || { tail -60 $O/ee_tests.txt; exit 1; }
  tail -15 $O/ee_tests.txt
  ;;
ee_ab)
  for shape in "1000 x 1 kb" "200 x 5 kb" "100k x 150 bp" "10 x 100 kb"; do
    for r in 0 1; do echo "== KWAGE_EE_REFINE=$r"; KWAGE_EE_REFINE=$r python tools/step_breakdown.py "$shape" 2>&1 | grep -E " ee "; done
  done > $O/r05_ee_step_breakdown_ab.txt 2>&1
  cat $O/r05_ee_step_breakdown_ab.txt
  ;;
ee_prof)
  cd /tmp; export TMPDIR=/tmp
  for shape in "1000 x 1 kb" "100k x 150 bp"; do
    tag=$(echo "$shape" | tr -d ' ')
    rm -rf $O/prof_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/tools/step_breakdown.py "$shape" > $O/ee_prof_$tag.txt 2>&1
    cp $(ls $O/prof_$tag/*/*kernel_stats.csv | tail -1) $O/r05_ee_${tag}_kernel_stats.csv
    rm -rf $O/prof_$tag
    grep -E " ee " $O/ee_prof_$tag.txt
    head -12 $O/r05_ee_${tag}_kernel_stats.csv | cut -c1-200
  done
  cd $R
  ;;
ee_stats)
  KWAGE_REFINE_STATIC=0 python tools/step_breakdown.py "1000 x 1 kb" 2>&1 | grep -E " ee "
  KWAGE_REFINE_STATIC=0 python tools/step_breakdown.py "100k x 150" 2>&1 | grep -E " ee "
  ;;
count_ab)
  for w in c2t c5s; do for r in 0 1; do
    KWAGE_EE_REFINE=$r python bench.py --workload $w --early-exit --no-cpu-baseline --no-sustained --also none --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w ee_refine=$r', l['ms_per_step'], l['roofline']['kernel'], l['roofline']['kernel_ms'], l['config']['hits_per_step'], l['result_check']['ok'])"
  done; done
  ;;
suite)
  python -m pytest tests -m gpu -x -q --durations=10 > $O/r05_gpu_suite_durations.txt 2>&1 || { tail -60 $O/r05_gpu_suite_durations.txt; exit 1; }
  tail -14 $O/r05_gpu_suite_durations.txt
  ;;
probe)
  df -h /tmp /dev/shm $R 2>&1; free -g; nproc; echo TMPDIR=$TMPDIR; cat /proc/cpuinfo | grep "model name" | sort | uniq -c
  ;;
identical)
  python tools/e2e_c2_identical.py --out $O/r05_c2_identical_db
  ;;
long1t)
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_node_cli.py tests/test_gpu_random_configs.py -m gpu -x -q -k "2_pow_20 or long_quer or segments or node_cli or random_config or walk_rows_many" > $O/long_tests.txt 2>&1 || { tail -40 $O/long_tests.txt; exit 1; }
  tail -3 $O/long_tests.txt
  python bench.py --workload long1t --no-cpu-baseline --also none --no-early-exit-block --steps 10 --warmup 3 2>/dev/null > $O/r05_long1t_bench.json
  python -c "
import json; l=json.loads(open('$O/r05_long1t_bench.json').read().strip().splitlines()[-1]); print('long1t', l['roofline']['kernel'], l['roofline']['kernel_ms'], l['roofline']['frac'], l['roofline']['frac_of_measured_stream'])"
  python tools/bench_long_query.py 2>&1 | tee $O/r05_long_query_segments.txt
  ;;
lines)
  python bench.py --steps 20 --warmup 5 > $O/r05_c2_bench.json 2> $O/r05_c2_bench.err
  for w in c2t c4 c5s c5 c3 c2q5kt c2q100kt; do python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 > $O/r05_${w}_bench.json 2>/dev/null; echo "line $w done"; done
  for w in c2 c2t c2q5k c3; do python bench.py --workload $w --early-exit --no-cpu-baseline --also none --steps 100 --warmup 5 > $O/r05_ee_${w}_bench.json 2>/dev/null; echo "ee line $w done"; done
  ;;
rocprof)
  cd /tmp; export TMPDIR=/tmp
  for w in c2 c2t c4 c5; do
    rm -rf $O/prof_$w
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 > $O/r05_${w}_bench_under_rocprof.json 2>/dev/null
    cp $(ls $O/prof_$w/*/*kernel_stats.csv | tail -1) $O/r05_${w}_kernel_stats.csv
    rm -rf $O/prof_$w; echo "rocprof $w done"
  done
  for w in c2 c2t c2q5k c3; do
    rm -rf $O/prof_ee_$w
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ee_$w -- python3 $R/bench.py --workload $w --early-exit --no-cpu-baseline --also none --steps 20 --warmup 5 > $O/r05_ee_${w}_bench_under_rocprof.json 2>/dev/null
    cp $(ls $O/prof_ee_$w/*/*kernel_stats.csv | tail -1) $O/r05_ee_${w}_kernel_stats.csv
    rm -rf $O/prof_ee_$w; echo "rocprof ee $w done"
  done
  cd $R
  ;;
pmc)
  python tools/pmc_refresh.py --round r05 c2 c2+bands c2t c3 c4 c5s c5 narrow narrowt long1t c2+ee c3+ee c2q5k+ee c2t+ee c2q5kt+ee c2q100kt+ee c4+ee c5s+ee c5+ee > $O/pmc_refresh.txt 2>&1 || true
  cp gpurun_out/pmc_r05/pmc_traffic.json gpurun_out/pmc_r05/r05_*_pmc_fetch_size.json $O/
  tail -20 $O/pmc_refresh.txt
  ;;
occupancy)
  python tools/pmc_occupancy.py --round r05 long1t c2 > $O/pmc_occupancy.txt 2>&1 || true
  cp gpurun_out/pmc_r05/r05_*_pmc_occupancy.json $O/ 2>/dev/null || true
  tail -12 $O/pmc_occupancy.txt
  ;;
sharded)
  for w in c2 c3 c5; do KWAGE_BENCH_FORCE_SHARDED=1 python bench.py --workload $w --no-cpu-baseline --also none --steps 10 --warmup 3 > $O/r05_${w}_bench_sharded_world1.json 2>/dev/null; echo "sharded $w done"; done
  KWAGE_BENCH_BACKEND=gloo KWAGE_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --also c3_strong --no-cpu-baseline --steps 10 --warmup 3 > $O/r05_c2_bench_two_ranks_one_gpu_gloo_also_c3_strong.json 2>$O/two_ranks.err || tail -20 $O/two_ranks.err
  python - <<PY
import json
l = json.loads(open("$O/r05_c2_bench_two_ranks_one_gpu_gloo_also_c3_strong.json").read().strip().splitlines()[-1])
b = l["also"]["c3_strong"]
print("2 ranks: c2", l["ms_per_step"], "c3_strong", b["ms_per_step"], b["scaling"], b["config"]["samples_per_gpu"], b["roofline"]["kernel"], b.get("exchange_check", {}).get("ok"), b["result_check"]["ok"])
PY
  ;;
proxy)
  python tools/strong_scaling_proxy.py --round r05 > $O/proxy.txt 2>&1
  cp gpurun_out/proxy_r05/r05_* $O/
  tail -12 $O/proxy.txt
  ;;
soak)
  python tools/soak_walk.py --launches 5000 --out $O/r05_soak.txt > /dev/null
  tail -5 $O/r05_soak.txt
  ;;
node)
  python tools/node_pipeline_stats.py > $O/r05_node_pipeline_stats.txt 2>&1
  tail -30 $O/r05_node_pipeline_stats.txt
  ;;
refine_knobs)
  for shape in "1000 x 1 kb" "100k x 150 bp"; do
    for seg in 32 64 128; do for un in 8 16; do
      echo "== KWAGE_REFINE_SEG_ROWS=$seg KWAGE_REFINE_UNROLL=$un"; KWAGE_REFINE_SEG_ROWS=$seg KWAGE_REFINE_UNROLL=$un python tools/step_breakdown.py "$shape" 2>&1 | grep -E " ee " | grep "t=1 "
    done; done
    for mg in 2 8; do echo "== KWAGE_REFINE_MAX_GROUPS=$mg"; KWAGE_REFINE_MAX_GROUPS=$mg python tools/step_breakdown.py "$shape" 2>&1 | grep -E " ee " | grep "t=1 "; done
    for wpc in 12 16; do echo "== KWAGE_SCREEN_WPC=$wpc"; KWAGE_SCREEN_WPC=$wpc python tools/step_breakdown.py "$shape" 2>&1 | grep -E " ee " | grep "t=1 "; done
  done > $O/r05_refine_knobs_ab.txt 2>&1
  cat $O/r05_refine_knobs_ab.txt | cut -c1-150
  ;;
count_check)
  for ck in 8 16 32 64; do for w in c2t c5s; do
    KWAGE_COUNT_SCREEN_CHECK=$ck python bench.py --workload $w --early-exit --no-cpu-baseline --no-sustained --also none --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w check=$ck', l['ms_per_step'], l['roofline']['kernel'], l['roofline']['kernel_ms'], l['result_check']['ok'])"
  done; done
  for ck in 8 16 32; do echo "check=$ck"; KWAGE_COUNT_SCREEN_CHECK=$ck python tools/step_breakdown.py "100k x 150" 2>&1 | grep " ee " | grep "t=0.8"; done
  ;;
walk_hint)
  for k in 1 2 4 8; do python bench.py --workload c2 --share-of $k --no-cpu-baseline --no-sustained --also none --no-early-exit-block --no-result-check --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 of $k', l['roofline']['kernel'], l['roofline']['kernel_ms'], l['roofline']['frac'], l['roofline']['frac_of_measured_stream'])"
  done
  ;;
two_ranks)
  KWAGE_BENCH_BACKEND=gloo KWAGE_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --also c3_strong --no-cpu-baseline --steps 10 --warmup 3 > $O/r05_c2_bench_two_ranks_one_gpu_gloo_also_c3_strong.json 2>$O/two_ranks.err || tail -20 $O/two_ranks.err
  ;;
trunc_ab)
  for shape in "200 x 5 kb" "10 x 100 kb" "1 x 10 kb"; do for tr in 0 1; do echo "== KWAGE_COUNT_TRUNC=$tr"; KWAGE_COUNT_TRUNC=$tr python tools/step_breakdown.py "$shape" 2>&1 | grep -E "t=0.8 "; done; done > $O/r05_count_trunc_ab.txt 2>&1
  cat $O/r05_count_trunc_ab.txt | cut -c1-230
  ;;
c5_groups)
  python tools/c5_ee_groups.py > $O/r05_c5_ee_groups.txt 2>&1
  cut -c1-300 $O/r05_c5_ee_groups.txt
  ;;
line)
  python bench.py --steps 20 --warmup 5 > $O/r05_c2_bench.json 2> $O/r05_c2_bench.err || { tail -30 $O/r05_c2_bench.err; exit 1; }
  python - <<PY
import json
l = json.loads(open("$O/r05_c2_bench.json").read().strip().splitlines()[-1])
print("c2", l["ms_per_step"], l["roofline"]["kernel"], l["roofline"]["frac"], json.dumps(l.get("early_exit"))[:1500])
c3 = l.get("also", {}).get("c3", {})
print("c3", c3.get("ms_per_step"), json.dumps(c3.get("early_exit"))[:1500])
PY
  ;;
esac
echo "stage $1 done"
