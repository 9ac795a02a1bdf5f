#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c11; mkdir -p $O
timeout 300 python tools/api_overhead.py > $O/api_overhead.txt 2>&1; cat $O/api_overhead.txt
