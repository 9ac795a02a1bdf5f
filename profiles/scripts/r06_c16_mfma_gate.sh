# round 6, call 16: DfMBackbone's gate on the matrix cores (bf16) vs the VALU kernel (DFM_GATE_VALU=1)
mkdir -p gpurun_out/c16
(python -m pytest tests/test_cost_gate_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/c16/tests.txt
python tools/gate_timing.py > gpurun_out/c16/gate_timing.txt 2>&1
DFM_GATE_VALU=1 python tools/gate_timing.py >> gpurun_out/c16/gate_timing.txt 2>&1
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2 3; do row backbone mfma_gate; DFM_GATE_VALU=1 row backbone valu_gate; done > gpurun_out/c16/rows.txt 2>&1
