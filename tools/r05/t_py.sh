#!/bin/bash
# usage: tools/r05/t_py.sh <script.py> [args]   (GPU box)
cd $GRAFT_REPO_ROOT
timeout 600 python "$@" 2>&1 | grep -v amdgpu.ids | tail -20
