#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "grouped_handoff or paired or split" 2>&1 | tail -15
