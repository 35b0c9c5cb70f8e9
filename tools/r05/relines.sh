#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/r05/lines.sh > gpurun_out/r05/refresh_lines.log 2>&1
