export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/r02g_pytest.log 2>&1; tail -4 gpurun_out/r02g_pytest.log
bash tools/ab_step.sh 2>&1 | tail -4
