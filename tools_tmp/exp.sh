run() { timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v,1) for k,v in d['kernel_ms'].items()})"; }
for nt in 64 128 256; do echo nt=$nt; TSFA_NT_0=$nt TSFA_NT_1=$nt TSFA_NT_2=$nt TSFA_NT_3=$nt TSFA_NT_5=$nt TSFA_NT_6=$nt run; done
