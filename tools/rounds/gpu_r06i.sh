#!/bin/bash
# r06i: forked capture after the _wgrad fix (tests), synthesis pipeline tests after the pinned PCM ring, default bench line with both graph variants
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_cli_gpu.py tests/test_vocoder_stft_gpu.py -x -q -m gpu -s ) 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|Error|assert|side_stream" | cut -c1-600 > gpurun_out/r06i_pytest.log; cat gpurun_out/r06i_pytest.log
timeout 900 python bench.py 2> gpurun_out/r06i_bench.err | tail -1 > gpurun_out/r06i_bench.json; python - <<'PY'
import json
l = json.load(open('gpurun_out/r06i_bench.json'))
c = l['config']
print(l['ms_per_step'], l['value'], c.get('hip_graph_ms_per_step'), c.get('hip_graph_forked_ms_per_step'), l['roofline']['frac'], c['synth'].get('rtf'), c['synth'].get('fp32_rtf'))
PY
