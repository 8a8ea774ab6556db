#!/bin/bash
# The multi-GPU launch line of the driver, at one rank, with the process group forced on: env parsing, nccl group, barrier, all-reduce MAX.
export HYPAD_BENCH_FORCE_DIST=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 \
  --no-cpu-baseline --no-scoring --no-drop-in --no-secondary 2> gpurun_out/launcher.err | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('launcher line ok: n_gpus', d['n_gpus'], 'ms_per_step %.3f' % d['ms_per_step'], 'value %.0f' % d['value'], 'scaling', d['scaling'])"
echo "rc=$?"; tail -n 3 gpurun_out/launcher.err
