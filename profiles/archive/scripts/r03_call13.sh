#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c13; mkdir -p $O
timeout 900 python -m pytest tests/test_plane_sweep_gpu.py tests/test_modules.py -q -m gpu -x -k "backward or grad or bwd or train or channels_last" 2>&1 | tail -4 | tee $O/tests.txt
timeout 200 python bench.py --workload backbone_train 2>/dev/null | tail -1 | tee $O/bench_train.json | cut -c1-300
timeout 200 python bench.py --workload backbone_train --reducer bucket 2>/dev/null | tail -1 | tee $O/bench_train_bucket.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('gradient_exchange'))"
timeout 200 python bench.py --workload backbone_train --reducer ddp 2>/dev/null | tail -1 | tee $O/bench_train_ddp.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('gradient_exchange'))"
