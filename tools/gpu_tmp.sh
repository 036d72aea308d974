export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_a_prodshape_gpu.py -q -x -k "vocoder or contraction" ) > gpurun_out/r02h_pytest_prod.log 2>&1; tail -5 gpurun_out/r02h_pytest_prod.log | cut -c1-600
( timeout 600 python -m pytest tests/test_vocoder_stft_gpu.py -q -x ) 2>&1 | tail -3
for i in 1 2; do ( cd _prev && python bench.py --mode synth 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-1 synth', d['ms_per_step'], d['value'])" ); python bench.py --mode synth 2>/dev/null | tail -1 | cut -c1-1700; done
