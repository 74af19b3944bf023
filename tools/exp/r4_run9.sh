#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r4_9
timeout 900 python bench.py > gpurun_out/r4_9/bench_line.json 2> gpurun_out/r4_9/bench.err; tail -3 gpurun_out/r4_9/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4_9/bench_line.json').read().strip().split('\n')[-1])
for k in ('value','ms_per_step','host_issue_ms_per_step','exact_f32','eager_f32'):
    print(k, d.get(k))
print('roofline', {k: d['roofline'].get(k) for k in ('avg_us','frac','traffic','avg_us_rocprof','frac_rocprof','dominant_kernel')})
print('fused', {k: d['roofline_fused'].get(k) for k in ('avg_us','frac','traffic','avg_us_rocprof')})
print('gate', d.get('gemm_gate'))
print('alt', {k: (v.get('value'), v.get('generator_us'), v.get('out_proj_us')) for k, v in d.get('alt_gemm', {}).items()})
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('gpu_over_cpu'), 'graph', d['config'].get('step_graph'))
P
timeout 600 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -3
