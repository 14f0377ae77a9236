#!/bin/bash
# workgroup-size sweep per family on long ragged shapes (TSFA_NT_<family index> hook of tsfa_api.cpp)
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()})"; }
for shape in 4096:8192 2049:4096; do
  echo "== ragged $shape base"; run --params efficient --n-series 5000 --ragged $shape
  for nt in 512 1024; do
    echo "== ragged $shape all families nt=$nt (trend 512)"
    TSFA_NT_0=$nt TSFA_NT_1=$nt TSFA_NT_2=$nt TSFA_NT_3=$nt TSFA_NT_5=$nt TSFA_NT_7=512 run --params efficient --n-series 5000 --ragged $shape
  done
done
