#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05; bash tools/r05/final.sh 3 > gpurun_out/r05/refresh3.log 2>&1
