export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_a_prodshape_gpu.py -q -s -k "full_size or contraction" ) > gpurun_out/r02e_pytest_prod.log 2>&1; grep -E "valid-frame|worst 8|median|passed|failed|^E " gpurun_out/r02e_pytest_prod.log | cut -c1-1200 | tail -12
timeout 300 python tools/bench_p.py ab FS2_P_ORDER=0 FS2_P_ORDER=1 > gpurun_out/r02e_order_ab.md 2>&1; cat gpurun_out/r02e_order_ab.md
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_a_prodshape_gpu.py ) > gpurun_out/r02e_pytest.log 2>&1; tail -5 gpurun_out/r02e_pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-fp32 > gpurun_out/r02e_bench.log 2>&1; tail -1 gpurun_out/r02e_bench.log | cut -c1-700
for side in 1 0; do
  rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof -o bench -- python bench.py --steps 6 --warmup 2 --windows 1 --side-stream $side --no-cpu-baseline --no-roofline --no-fp32 > gpurun_out/prof.log 2>&1
  DB=$(find gpurun_out/prof -name '*.db' | head -1)
  python tools/rocpd_summary.py $DB 10 shapes > gpurun_out/r02e_kernel_trace_side${side}.md 2>&1
done
rm -rf gpurun_out/prof
sed -n '/## steady state/,$p' gpurun_out/r02e_kernel_trace_side1.md | head -50
