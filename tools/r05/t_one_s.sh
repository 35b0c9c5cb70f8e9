#!/bin/bash
# usage: tools/r05/t_one_s.sh <pytest args...>   (GPU box; prints of the tests kept)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest "$@" -q -x -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/t_one.log
