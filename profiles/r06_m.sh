#!/bin/bash
# k_general after the explicit vmcnt wait in the one-wavefront barrier; then the whole gpu suite
O=gpurun_out/r06m; mkdir -p $O
python profiles/dbg_cwt_tmp.py > $O/dbg.log 2>&1; cat $O/dbg.log
timeout 1500 python -m pytest tests/test_param_beyond.py -q -m gpu > $O/beyond.log 2>&1; tail -5 $O/beyond.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
