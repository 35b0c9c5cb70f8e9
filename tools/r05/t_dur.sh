#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -x -m gpu --durations=25 2>&1 | tail -40 > gpurun_out/t_dur.log
