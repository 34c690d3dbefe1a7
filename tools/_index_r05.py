#!/usr/bin/env python3
"""(one-off, round 5) the r05 entries of profiles/INDEX.json"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, "profiles", "INDEX.json")
idx = json.load(open(p))
idx["r05"] = [
 {"files": "`r05_c2_bench.json` (+ `c2t`, `c3`, `c4`, `c5s`, `c5`)",
  "what": "the `bench.py` line per workload on the FINAL code of round 5. `r05_c2_bench.json` is the driver's default command: the C2 headline with `sustained`, `result_check`, its **`early_exit` block** (the same batch on the same resident matrix with `KWAGE_SEARCH_EARLY_EXIT`: what `kwage` and `kwage_node` run by default), **`cpu_baseline` = the reference `kwage` on the IDENTICAL database** (the resident matrix read back from HBM and written as 49 reference-format `.db` files, 104.9 GB, whole hit list compared with the timed kernel's; plus the bounded no-early-exit sample) AND the `also.c3` block with its own `early_exit`; the others 10 steps, `--no-cpu-baseline`. Every line carries the box it was taken on (`config.box`: `kwage_device_fingerprint`). C5 with BASELINE.md's 10 k x 1 kb queries",
  "command": "`bash tools/r05_campaign.sh lines` (= `python bench.py --steps 20 --warmup 5`; `python bench.py --workload w --no-cpu-baseline --steps 10 --warmup 3`)"},
 {"files": "`r05_ee_c2_bench.json`, `r05_ee_c2t_bench.json`, `r05_ee_c2q5k_bench.json`, `r05_ee_c3_bench.json`",
  "what": "WHOLE runs with `--early-exit` (every timed step searches with the reference's early exit: the commands behind the `@ee` PMC passes): C2, C2 at t = 0.8 (count path), 200 x 5 kb queries against C2's matrix, C3. `roofline.frac` here is the batch's ALGORITHMIC bytes over the early-exit kernel time (bytes mostly not read): the HBM figure is `traffic` / `frac_of_fetched` in TABLES.md's early-exit table",
  "command": "`python bench.py --workload w --early-exit --no-cpu-baseline --also none --steps 100 --warmup 5`"},
 {"files": "`r05_c2_kernel_stats.csv`, `r05_c2_bench_under_rocprof.json` (+ `c2t`, `c4`, `c5`; `ee_c2`, `ee_c2t`, `ee_c2q5k`, `ee_c3`)",
  "what": "`rocprofv3 --kernel-trace --stats` summaries of the same commands and the line each profiled run printed: the default command (C2's walk kernel, its `early_exit` block's `and_screen_kernel` / `and_refine_kernel` / `and_refine_emit_kernel`, C3's tiled kernel and C3's early-exit stage), C2 at t = 0.8 (`count_walk_kernel` + `count_screen_kernel` / `count_refine_kernel` / `count_refine_emit_kernel`), the C4 and C5 shares, and the four `--early-exit` runs (per-launch averages of the stage's three kernels)",
  "command": "`bash tools/r05_campaign.sh rocprof` (`rocprofv3 --kernel-trace --stats --output-format csv -d … -- python3 bench.py --workload w [--early-exit --also none] --no-cpu-baseline --steps 20 --warmup 5`)"},
 {"files": "`r05_*_pmc_fetch_size.json`, `pmc_traffic.json`",
  "what": "HBM bytes per launch from separate `rocprofv3 --pmc FETCH_SIZE` passes (factor calibrated in each pass on `stream_read_kernel`), bound to kernel name + code hash: the nominal kernels of C2 (walk and band form), C2t, C3, C4, C5s, C5, narrow, narrowt, long1t, and -- new -- the EARLY-EXIT stages `c2@ee`, `c3@ee`, `c2q5k@ee`, `c2t@ee` (bytes of all the stage's launches per step, per kernel in `per_kernel_bytes_per_step`): what `early_exit.fetched_bytes` / `frac_of_fetched` of the lines quote",
  "command": "`bash tools/r05_campaign.sh pmc` (`python tools/pmc_refresh.py --round r05 c2 c2+bands c2t c3 c4 c5s c5 narrow narrowt long1t c2+ee c3+ee c2q5k+ee c2t+ee`)"},
 {"files": "`r05_long1t_pmc_occupancy.json`, `r05_c2_pmc_occupancy.json`",
  "what": "SQ / GRBM counter passes: one 100 kb query at t = 0.9 after the 7-plane block counters (share of wave time issuing instructions, against round 4's 42 %) and C2's walk kernel",
  "command": "`bash tools/r05_campaign.sh occupancy` (`python tools/pmc_occupancy.py --round r05 long1t c2`)"},
 {"files": "`r05_c2_identical_db.txt`, `r05_c2_identical_db.json`",
  "what": "BOTH BINARIES ON IDENTICAL FILES AT C2'S SIZE (BASELINE.md section 4): the resident C2 matrix read back from HBM and written as 49 reference-format `.db` files (2^23 rows x <= 2048 columns, 104.9 GB, in /dev/shm of the box: 256 CPUs, 3 TB RAM) + the batch's 1 000 x 1 kb queries as FASTA; `oracle/_ref/kwage` (OpenMP, 49 threads, page cache warm, best of 2) and `kwage_amd/bin/kwage` at -t 1.0 and -t 0.8: walls, load / search split, **reports identical: True** (whole lists, per query), and the same command with the whole database loaded (`KWAGE_SPARSE=0`)",
  "command": "`bash tools/r05_campaign.sh identical` (`python tools/e2e_c2_identical.py`)"},
 {"files": "`r05_ee_step_breakdown_ab.txt`, `r05_ee_step_breakdown_ab_t1.txt`, `r05_ee_1000x1kb_kernel_stats.csv`, `r05_ee_100kx150bp_kernel_stats.csv`",
  "what": "the early-exit path before / after in ONE process per shape (knob `ee_refine` = 0: the tiled kernels of round 4; 1: screen + refine): 1000 x 1 kb, 200 x 5 kb, 100 k x 150 bp, 10 x 100 kb against 100 k samples x 2^20 rows, at t = 1 and t = 0.8 (`_t1`: the first measurement of the t = 1 form, before the count path existed); rocprofv3 per-kernel averages of the stage's three launches on two of the shapes",
  "command": "`bash tools/r05_campaign.sh ee_ab` / `ee_prof` (`KWAGE_EE_REFINE=r python tools/step_breakdown.py \"shape\"`)"},
 {"files": "`r05_long_query_segments.txt`, `r05_long1t_bench.json`",
  "what": "few long queries (1 x 100 kb, 4 x 1 Mb) through segments / the persistent count kernel after the block counters, and the `long1t` line: 2 % over round 4's file -- the launch's fixed part, not the adders, caps one 100 kb query (DESIGN 3.3)",
  "command": "`bash tools/r05_campaign.sh long1t` (`python tools/bench_long_query.py`; `python bench.py --workload long1t …`)"},
 {"files": "`r05_c2_bench_sharded_world1.json`, `r05_c3_bench_sharded_world1.json`, `r05_c5_bench_sharded_world1.json`, `r05_c2_bench_two_ranks_one_gpu_gloo_also_c3_strong.json`",
  "what": "the multi-GPU code path on one GPU: one rank over RCCL (`KWAGE_BENCH_FORCE_SHARDED=1`) for C2, C3, C5, and TWO ranks sharing the device over gloo running the driver's N > 1 command with `--also c3_strong`: the headline (weak) plus C3's columns split over the two ranks (the fixed-total-work block), each with `exchange_check` and `result_check`",
  "command": "`bash tools/r05_campaign.sh sharded`"},
 {"files": "`r05_strong_scaling_proxy.json`, `r05_proxy_<workload>_of<K>_kernel_stats.csv`",
  "what": "the one-GPU proxy of the strong-scaling curve re-taken on round 5's code (hash in the file): C3 and C2 columns split 1 / 2 / 4 / 8 ways, each share in a fresh process under rocprofv3",
  "command": "`bash tools/r05_campaign.sh proxy` (`python tools/strong_scaling_proxy.py --round r05`)"},
 {"files": "`r05_soak.txt`",
  "what": "the soak of the cut-pair protocols re-taken on round 5's code (the hash `bench.py` prints is in its header): tens of thousands of launches of the three persistent kernels at wave counts {5, 3001, 16384, 30000}, every list compared, every exchange buffer read back all zero",
  "command": "`bash tools/r05_campaign.sh soak` (`python tools/soak_walk.py --launches 5000`)"},
 {"files": "`r05_node_pipeline_stats.txt`",
  "what": "`kwage_node` on the five-group test-scale database: one rank over RCCL, two and three rehearsed ranks, and -- new -- the same database with a budget of half / a quarter of it per rank (`KWAGE_MAX_GROUP_BYTES`): passes, loading per pass, what the searches waited for the rank's own query parser; reports identical to `kwage`'s in every run",
  "command": "`bash tools/r05_campaign.sh node` (`python tools/node_pipeline_stats.py`)"},
 {"files": "`r05_gpu_suite_durations.txt`",
  "what": "`pytest -m gpu --durations=10` on the final library",
  "command": "`bash tools/r05_campaign.sh suite`"},
 {"files": "`r05_prune.txt`",
  "what": "template axes and knobs removed in round 5: instantiation counts per kernel (325 -> 223 with 48 new ones aboard), library size, compile wall, the knobs and the A/B files that retired them",
  "command": "`nm -C kwage_amd/lib/engine.o | grep -c __device_stub__`; `ls -l kwage_amd/lib/libkwage_amd.so`; `time make -C kwage_amd/csrc ../lib/engine.o`"},
]
json.dump(idx, open(p, "w"), indent=1, ensure_ascii=False)
print("r05 entries:", len(idx["r05"]))
