#!/bin/bash
# r06d: residual-branch gradients added by the LayerNorm backward (fs2_ln_bwd_sum) instead of the dgrad epilogues - parity, then step A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_a_prodshape_gpu.py tests/test_model_gpu.py tests/test_libritts_shape_gpu.py tests/test_graph_gpu.py tests/test_z_bf16_budget_gpu.py -x -q -m gpu -s ) 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|Error|assert|ratios|valid-frame|rel-Frobenius" | cut -c1-900 > gpurun_out/r06d_pytest.log; cat gpurun_out/r06d_pytest.log
bash tools/ab_lib.sh 3 | tee gpurun_out/r06d_ab_step.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d_ab_res_in_ln.log
# same library, the engine switch both ways, alternating (one process: the model is rebuilt per setting)
import sys, time, json; sys.path.insert(0, '.')
import fastspeech2_amd; fastspeech2_amd.configure_hw_queues()
import torch, bench
args = bench.parse([])
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
model, loss_fn, opt, b, pcfg, mcfg = bench.build(args, dev, 0, 1)
step, _ = bench.make_step(model, loss_fn, opt, b, None)
for rnd in range(3):
    for mode in (False, True):
        model._engine.res_in_ln = mode
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize()
        print(f"round {rnd} res_in_ln={mode}: {(time.perf_counter()-t0)/30*1e3:.3f} ms/step", flush=True)
PY
