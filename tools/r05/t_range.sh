#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -m gpu -k "range_safe or nonfinite or gradients_and_adam or clipping" -s 2>&1 | tail -60 > gpurun_out/t_range.log
