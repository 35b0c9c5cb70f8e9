#!/bin/bash
python3 tools/r04/concurrent_bwd.py
for cfg in "160 96" "144 112" "128 128" "176 80" "192 64" "128 256" "160 192"; do set -- $cfg; RNERF_DGRAD_WG=$1 RNERF_WGRAD_WGS=$2 python3 tools/r04/concurrent_bwd.py; done
python3 tools/r04/concurrent_bwd.py
