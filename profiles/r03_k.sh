#!/bin/bash
# round 3, after the register-tile rewrite of k_ar's lagged sums and k_cwtpeaks' phase A: gpu tests + headline bench line
TAG=${1:-r03_k}
export TMPDIR=/tmp
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --no-e2e > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; echo
TSFA_LIB=tsfresh_amd/libtsfresh_amd_ticks.so python profiles/phase_ticks.py > $O/phase_ticks.md 2>&1; grep -E "ar|cwt" $O/phase_ticks.md
