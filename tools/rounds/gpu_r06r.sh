#!/bin/bash
# r06r: default bench line + LibriTTS-shaped line on the current tree
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py 2> gpurun_out/r06r_bench.err | tail -1 > gpurun_out/r06r_bench.json
timeout 900 python bench.py --workload libritts --no-cpu-baseline --no-synth --no-fp32 2> gpurun_out/r06r_bench_libritts.err | tail -1 > gpurun_out/r06r_bench_libritts.json
python - <<'PY'
import json
for f in ('gpurun_out/r06r_bench.json', 'gpurun_out/r06r_bench_libritts.json'):
    l = json.load(open(f)); c = l['config']; r = l['roofline'] or {}
    print(f, l['ms_per_step'], l['value'], 'graph', c.get('hip_graph_ms_per_step'), c.get('hip_graph_forked_ms_per_step'), 'frac', r.get('frac'), 'step_frac', r.get('step_frac_of_peak'), 'valid', c.get('valid_row_fraction'), (c.get('synth') or {}).get('rtf'), (c.get('synth') or {}).get('fp32_rtf'))
PY
