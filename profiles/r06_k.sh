#!/bin/bash
# side-lane combinations (TSFA_PAIR) and stream counts at the headline and configs[3] shapes
O=gpurun_out/r06k; mkdir -p $O
run() { # name, env..., -- args
  name=$1; shift
  env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e $ARGS > $O/$name.json 2>$O/$name.err
  python - <<PY
import json
d=json.loads(open('$O/$name.json').read().strip().split('\n')[-1])
print('$name', round(d['ms_per_step'],3), round(sum(d.get('kernel_ms',{}).values()),3))
PY
}
for shape in "100000 1024" "125000 256"; do
  set -- $shape
  ARGS="--n-series $1 --length $2 --params comprehensive"
  echo "== $shape"
  run base_$2 X=1
  run seq_$2 TSFA_PAIR=seq
  run seqcwt_$2 TSFA_PAIR=seq,cwt
  run seqcwtspec_$2 TSFA_PAIR=seq,cwt,spectral
  run seqar_$2 TSFA_PAIR=seq,ar
  run all_$2 TSFA_PAIR=seq,cwt,spectral,ar,trend
  run streams2_$2 TSFA_STREAMS=2
  run streams3_$2 TSFA_STREAMS=3
done
