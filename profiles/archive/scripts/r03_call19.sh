#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_sweep_conv_gpu.py -q -m gpu 2>&1 | tail -6
