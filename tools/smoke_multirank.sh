#!/bin/bash
# Rehearsal of bench.py's N=2 control flow on a one-GPU box (gloo, both ranks on cuda:0) + check of the gathered frame
# against a single-rank render of the same frame.  Diagnostic helper; not a measurement.
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=${N:-2}; read W H <<< $(python3 -c "import math; n=int('$N'); s=math.sqrt(n); print(8*int(round(1920*s/8.0)), 8*int(round(1080*s/8.0)))")
EU_BENCH_SMOKE_GLOO=1 EU_BENCH_DUMP=/tmp/eu_two.npy python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 2 --warmup 1 --no-cpu-baseline | tail -1 | cut -c1-400
EU_BENCH_DUMP=/tmp/eu_one.npy python bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --width $W --height $H | tail -1 | cut -c1-200
python - <<'PY'
import numpy as np
a=np.load("/tmp/eu_two.npy"); b=np.load("/tmp/eu_one.npy")
print("two-rank frame == one-rank frame:", a.shape, b.shape, bool(np.array_equal(a,b)))
PY
