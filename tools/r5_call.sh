mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_final.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/pytest_final.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/pytest_final.log | head
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_final.log 2> gpurun_out/bench_final.err; echo "bench exit $?"; tail -c 600 gpurun_out/bench_final.log
