#!/bin/bash
# round 4, call I: marginal cost of k_entropy_bits' phases -- the headline step with one phase left out at a time (timing
# only: the results of those builds are wrong by construction)
export TMPDIR=/tmp
O=gpurun_out/r04_i; rm -rf $O; mkdir -p $O
{
for lib in libtsfresh_amd.so; do
  TSFA_LIB=$PWD/tsfresh_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-e2e --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-40s step %.2f ms  k_entropy %.2f  k_sort %.2f  parity %s' % ('$lib', d['ms_per_step'], d['kernel_ms']['k_entropy'], d['kernel_ms']['k_sort'], str(d.get('parity_sample'))[:20]))"
done
} > $O/entropy_phase_cost2.txt 2>&1
cat $O/entropy_phase_cost2.txt
