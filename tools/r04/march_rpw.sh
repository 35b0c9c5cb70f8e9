#!/bin/bash
# usage (GPU box): bash tools/r04/march_rpw.sh  -> march_kernel at 4096 rays x 1536 nodes with 16 / 8 / 4 rays per workgroup
for r in 16 8 4; do echo "== rays per workgroup $r"; RNERF_MARCH_RPW=$r python3 tools/march_time.py reference 2>/dev/null | grep -v refractive; done
