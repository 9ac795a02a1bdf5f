#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c18; mkdir -p $O
timeout 200 python tools/store_probe_sweep.py 2>&1 | tail -7 | tee $O/store_probe.txt
