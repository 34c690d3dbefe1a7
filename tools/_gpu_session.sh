set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout -k 10 300 python tools/walk_sizes.py 2048 3072 > gpurun_out/r2d/walk_sizes.txt 2>&1; tail -20 gpurun_out/r2d/walk_sizes.txt
timeout -k 10 1000 python -m pytest tests/test_cli_streaming.py tests/test_fuzz_vs_reference_binary.py tests/test_gpu_parity.py tests/test_gpu_random_configs.py -x -q -m gpu -k "cli or stream or batches or fuzz or walk or random_conf" > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2d/pytest.log
tail -30 gpurun_out/r2d/pytest.log
