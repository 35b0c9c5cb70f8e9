#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -m gpu -k "noise_std or mask_bbox or default_flags" -s 2>&1 | tail -40 > gpurun_out/t_noise.log
