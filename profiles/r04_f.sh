#!/bin/bash
# round 4, call F: phase clocks (diagnostics build), long-series entropy table, the gpu tests touched since call E
export TMPDIR=/tmp
O=gpurun_out/r04_f; rm -rf $O; mkdir -p $O
TSFA_LIB=$PWD/tsfresh_amd/libtsfresh_amd_ticks.so timeout 600 python profiles/phase_ticks.py > $O/phase_ticks.md 2> $O/phase_ticks.err; tail -70 $O/phase_ticks.md
timeout 1500 python profiles/long_entropy.py > $O/long_entropy.md 2> $O/long_entropy.err; cat $O/long_entropy.md
TSFA_PARITY_SKIPS_MD=$O/parity_skips.md timeout 900 python -m pytest tests -m gpu -q -x -k "every_cell or offset_fuzz or stuck_sensors or seq" > $O/pytest_some.log 2>&1; tail -3 $O/pytest_some.log
