#!/bin/bash
# The measurement campaign behind profiles/r04_*: run a stage on the GPU box, everything lands under gpurun_out/r04/ with
# its final name (copy into profiles/ afterwards).   bash tools/r04_campaign.sh lines|rocprof|pmc|sharded|proxy|soak|node|suite
set -e
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
case "$1" in
lines)
  python bench.py --steps 20 --warmup 5 > $O/r04_c2_bench.json 2> $O/r04_c2_bench.err
  for w in c2t c4 c5s c5; do python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 > $O/r04_${w}_bench.json 2>/dev/null; done
  python bench.py --workload c3 --no-cpu-baseline --steps 10 --warmup 3 > $O/r04_c3_bench.json 2>/dev/null
  ;;
rocprof)
  cd /tmp; export TMPDIR=/tmp
  for w in c2 c2t c4 c5; do
    rm -rf $O/prof_$w
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 > $O/r04_${w}_bench_under_rocprof.json 2>/dev/null
    cp $(ls $O/prof_$w/*/*kernel_stats.csv | tail -1) $O/r04_${w}_kernel_stats.csv
    rm -rf $O/prof_$w
  done
  ;;
pmc)
  python tools/pmc_refresh.py --round r04 c2 c2+bands c2t c3 c4 c5s c5 narrow narrowt long1t > $O/pmc_refresh.txt 2>&1
  cp gpurun_out/pmc_r04/pmc_traffic.json gpurun_out/pmc_r04/r04_*_pmc_fetch_size.json $O/
  ;;
sharded)
  for w in c2 c3 c5; do KWAGE_BENCH_FORCE_SHARDED=1 python bench.py --workload $w --no-cpu-baseline --also none --steps 10 --warmup 3 > $O/r04_${w}_bench_sharded_world1.json 2>/dev/null; done
  KWAGE_BENCH_BACKEND=gloo KWAGE_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --also none --steps 10 --warmup 3 > $O/r04_c2_bench_two_ranks_one_gpu_gloo.json 2>/dev/null
  KWAGE_BENCH_BACKEND=gloo KWAGE_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --scaling strong --also none --steps 10 --warmup 3 > $O/r04_c2_bench_two_ranks_one_gpu_gloo_strong.json 2>/dev/null
  ;;
proxy)
  python tools/strong_scaling_proxy.py --round r04 > $O/proxy.txt 2>&1
  cp gpurun_out/proxy_r04/r04_* $O/
  ;;
soak)
  python tools/soak_walk.py --launches 5000 --out $O/r04_soak.txt > /dev/null
  ;;
node)
  python tools/node_pipeline_stats.py > $O/r04_node_pipeline_stats.txt 2>&1
  ;;
suite)
  python -m pytest tests -m gpu -x -q --durations=10 > $O/r04_gpu_suite_durations.txt 2>&1
  ;;
esac
echo "stage $1 done"; ls $O | tail -40
