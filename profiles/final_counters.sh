export TMPDIR=/tmp
bash profiles/pmc_issue.sh > gpurun_out/pmc_issue_f.log 2>&1; cp gpurun_out/pmc_issue/summary.md gpurun_out/r02_f_pmc_issue.md
bash profiles/pmc_entropy.sh > gpurun_out/r02_f_pmc_entropy.txt 2>&1
python profiles/phase_ticks.py > gpurun_out/r02_f_phase_ticks.md 2>&1
bash profiles/shapes.sh > /dev/null 2>&1; cp gpurun_out/exp/shapes.txt gpurun_out/r02_f_shapes.txt
tail -3 gpurun_out/r02_f_pmc_entropy.txt; cat gpurun_out/r02_f_shapes.txt; head -14 gpurun_out/r02_f_pmc_issue.md
