#!/bin/bash
# development aid: bench.py at the small configs with the automatic and the PCG solver choice
for c in 1 2; do for sv in 0 2; do
  python bench.py --config $c --solver $sv --steps 10 --warmup 3 --cpu-iters 0 2>/dev/null | grep "^{" | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c solver $sv', d['value'], 'it/s', d['ms_per_step'], 'ms', d['lm']['stage_ms_per_step'], 'lin', d['lm']['linear_iterations'])"
done; done
