#!/bin/bash
# usage: tools/r05/t_one.sh <pytest args...>   (GPU box)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest "$@" -q -x -m gpu 2>&1 | tail -30 > gpurun_out/t_one.log
