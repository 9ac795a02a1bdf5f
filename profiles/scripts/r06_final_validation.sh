# round 6: validation of the final tree -- full GPU suite, smoke(), the default bench line, rocprofv3 kernel statistics
OUT=gpurun_out/final; mkdir -p $OUT
(python -m pytest tests -x -q -m gpu 2>&1 | tail -5) > $OUT/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tools/kernel_stats.sh $GRAFT_REPO_ROOT/$OUT/ks default:"--no-secondary --no-cpu-baseline --no-traffic" backbone:"--workload backbone --steps 10 --warmup 3" backbone_train:"--workload backbone_train --steps 5 --warmup 2" stereo_train:"--workload stereo_train --steps 5 --warmup 2" neck:"--workload neck --steps 10 --warmup 3"
