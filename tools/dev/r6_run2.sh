#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest2.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r06/gputest2.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench1.json 2> gpurun_out/r06/bench1.err; echo "bench rc=$?"
python -c "import json;d=json.load(open('gpurun_out/r06/bench1.json'));print(d['ms_per_step'],d['roofline']['frac'],d['roofline']['frac_pipeline'],d.get('steady'),d['c3_16384']['ms_per_image'],d['c3_16384']['roofline']['frac'])"
