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
