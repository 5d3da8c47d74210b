#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== 2 ranks on one GPU (gloo): the multi-rank code path of bench.py"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --dist-backend gloo > gpurun_out/bench_2rank.log 2>&1; echo "exit $?"
tail -1 gpurun_out/bench_2rank.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'n_gpus', 'ms_per_step', 'scaling')})
print({k: (v if not isinstance(v, float) else round(v, 3)) for k, v in d['train_dp'].items() if k != 'roofline'})
" || tail -20 gpurun_out/bench_2rank.log
