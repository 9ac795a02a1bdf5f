#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c37; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3d_g_gpu.py tests/test_modules.py tests/test_config_build.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/train_step_timing.py 2>&1 | grep -v "MIOpen(HIP)\|amdgpu.ids" | tee $O/train_step.txt
for W in backbone_train backbone; do timeout 200 python bench.py --workload $W --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_$W.json; cat $O/bench_$W.json; done
