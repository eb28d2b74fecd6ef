#!/bin/bash
# the headline loop without parity / CPU legs: value, ms per step, per-kernel launch times (for A/B runs on the GPU box)
python bench.py --no-parity --no-cpu-baseline --no-csr > gpurun_out/b.log 2>&1; tail -c 300 gpurun_out/b.log | head -5
grep "^{" gpurun_out/b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [(k['kernel'], round(k['avg_launch_ms'],4), round(k['frac_of_8000'],3)) for k in d['step_kernels']], d.get('batched_25_steps_per_sync_iters_per_sec'), d['roofline'].get('back_to_back_ms', d['roofline'].get('spmv_alone_back_to_back_ms')))"
