# round 6, call 15: counter bytes for secondary rows (the harness's --mode bwd / nhwc), then the default bench line
mkdir -p gpurun_out/c15
(tools/sweep_bench --workload nstar --mode bwd --rounds 3 --launches 2; tools/sweep_bench --workload kitti --mode bwd --rounds 3 --launches 3; tools/sweep_bench --workload kitti --mode nhwc --rounds 3 --launches 3) > gpurun_out/c15/harness_modes.txt 2>&1
python bench.py > gpurun_out/c15/bench_default.json 2> gpurun_out/c15/bench_default.err
