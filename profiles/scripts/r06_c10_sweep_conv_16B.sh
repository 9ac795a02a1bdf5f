# round 6, call 10: 16-byte stores in sweep_conv_kernel's epilogue (v_permlane16_swap between channel groups)
mkdir -p gpurun_out/c10
(python -m pytest tests/test_sweep_conv_gpu.py tests/test_modules.py tests/test_path_parity_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/c10/tests.txt
python tools/sweep_conv_timing.py > gpurun_out/c10/sweep_conv_timing.txt 2>&1
row() { DFM_FEATS_NHWC=1 python bench.py --workload $1 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['ms_per_step'], l['roofline']['frac'])"; }
for wl in backbone stereo_infer backbone stereo_infer; do row $wl; done > gpurun_out/c10/rows.txt 2>&1
