#!/bin/bash
# round 4, call A: the new gpu tests, the issue-rate microbenchmark, the bench line of the tree as it stands
export TMPDIR=/tmp
O=gpurun_out/r04_a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "bench_self_launch or alternating_langevin" > $O/pytest_new.log 2>&1; tail -3 $O/pytest_new.log
timeout 300 profiles/lab/build/issue_rates > $O/issue_rates.jsonl 2> $O/issue_rates.err; wc -l $O/issue_rates.jsonl; tail -2 $O/issue_rates.err
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
