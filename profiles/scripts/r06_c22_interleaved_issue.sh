# round 6, call 22: DfMBackbone's two stacks issued alternately (layer by layer) vs the whole mono stack first
mkdir -p gpurun_out/c22
(python -m pytest tests/test_modules.py tests/test_path_parity_gpu.py tests/test_fast_path.py tests/test_conv3d_to1n_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/c22/tests.txt
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', '$1', l['ms_per_step'], l['roofline']['frac'])"; }
for i in 1 2 3; do row backbone interleaved; DFM_BACKBONE_SEQUENTIAL_ISSUE=1 row backbone sequential; done > gpurun_out/c22/rows.txt 2>&1
for i in 1 2; do row stereo_infer interleaved; DFM_BACKBONE_SEQUENTIAL_ISSUE=1 row stereo_infer sequential; done >> gpurun_out/c22/rows.txt 2>&1
python tools/host_overhead_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c22/host_overhead.txt
