mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_r5a.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_r5a.log
timeout 900 python bench.py > gpurun_out/bench_r5a.log 2> gpurun_out/bench_r5a.err; echo "bench exit $?"; tail -c 3000 gpurun_out/bench_r5a.log; tail -5 gpurun_out/bench_r5a.err
