#!/bin/bash
# round 6, GPU call 1: the new stage tests + whole GPU suite, a baseline bench line, the blur kernel's counters, the issue-rate ubenches
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r06/gputest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench0.json 2> gpurun_out/r06/bench0.err; echo "bench rc=$?"
python -c "import json;d=json.load(open('gpurun_out/r06/bench0.json'));print(d['ms_per_step'],d['roofline']['frac'],d['roofline']['frac_pipeline'],d.get('steady'))"
bash tools/dev/pmc_blur.sh > gpurun_out/r06/pmc_blur_team_kernel.txt 2>&1
./tools/ubench/valu_rate_bench > gpurun_out/r06/valu_issue_rate.txt 2>&1
./tools/ubench/valu2_bench > gpurun_out/r06/valu2.txt 2>&1
tail -20 gpurun_out/r06/pmc_blur_team_kernel.txt
