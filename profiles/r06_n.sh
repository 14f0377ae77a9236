#!/bin/bash
# whole gpu suite of the build with the explicit global-memory wait in the long build's barrier; phase ticks at three shapes
O=gpurun_out/r06n; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
export TSFA_LIB=$PWD/tsfresh_amd/libtsfresh_amd_ticks.so
python profiles/phase_ticks.py --n-series 20000 --length 1024 > $O/ticks_1024.md 2>$O/ticks_1024.err
python profiles/phase_ticks.py --n-series 30000 --length 256 > $O/ticks_256.md 2>$O/ticks_256.err
python profiles/phase_ticks.py --n-series 2000 --ragged 4096:8192 --params efficient > $O/ticks_cfg4.md 2>$O/ticks_cfg4.err
head -30 $O/ticks_cfg4.md
