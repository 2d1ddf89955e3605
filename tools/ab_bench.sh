#!/bin/bash
# A/B several builds of libfsaempc on ONE device, interleaved rounds (guide rule 24).
# usage: tools/ab_bench.sh ROUNDS "bench args" lib1.so lib2.so ...
R=$1; shift; ARGS=$1; shift
for i in $(seq 1 $R); do
  for L in "$@"; do
    FSAEMPC_LIB=$L python bench.py --steps 5 --warmup 1 --no-cpu-baseline $ARGS 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], 'round $i', '%.0f QP/s' % d['value'], 'kernel %.2f ms' % d['config']['solve_kernel_ms'], 'iters %.2f' % d['config']['mean_ipm_iterations'])"
  done
done
