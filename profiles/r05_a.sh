#!/bin/bash
# round 5, call A (first thing in the round): the 16 gpu tests that had never run on a device (parameter sweep on the
# degenerate / offset / long sets, ADF autolag BIC / t-stat / None, stuck-sensor AR(16+)), then the three counter
# passes on the SAME build (issue, HBM traffic, instruction cache).
export TMPDIR=/tmp
O=gpurun_out/r05_a; rm -rf $O; mkdir -p $O
sha256sum tsfresh_amd/libtsfresh_amd.so | cut -c1-16 > $O/lib_sha16.txt
TSFA_TEST_NEW_ON_HARDWARE=1 TSFA_ADF_AUTOLAG=1 timeout 300 python -m pytest tests/test_param_sweep.py tests/test_adf_autolag.py tests/test_ar_stuck.py -m gpu -q -rs > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log; tail -25 $O/pytest_new.log
bash profiles/pmc_issue.sh > $O/pmc_issue.log 2>&1; cp gpurun_out/pmc_issue/summary.md $O/pmc_issue.md; cp gpurun_out/pmc_issue/valu_issue.json $O/valu_issue.json
bash profiles/pmc_hbm.sh > $O/pmc_hbm.log 2>&1; cp gpurun_out/hbm/traffic.json $O/hbm_traffic.json
bash profiles/pmc_icache.sh > $O/pmc_icache.log 2>&1; cp gpurun_out/pmc_icache/summary.md $O/pmc_icache.md
tail -20 $O/pmc_icache.md
