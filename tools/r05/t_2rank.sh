#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train_2rank.py -q -x -m gpu -s 2>&1 | tail -30 > gpurun_out/t_2rank.log
