"""Dev tool: the fields of a bench.py JSON line that the round's experiments compare."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    r, f, g = d.get('roofline', {}), d.get('roofline_fused', {}), d.get('gemm_gate', {})
    print('%s value %.1f ms/step %.4f host %.3f | sampler %.1f us frac %s | fused %.1f us frac %s traffic %s | gen %.1f out %.1f | checksum %s' % (
        sys.argv[1] if len(sys.argv) > 1 else '', d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step', 0), r.get('avg_us', 0), r.get('frac'),
        f.get('avg_us', 0), f.get('frac'), f.get('traffic'), g.get('generator_us', 0), g.get('out_proj_us', 0), d.get('config', {}).get('checksum')))
