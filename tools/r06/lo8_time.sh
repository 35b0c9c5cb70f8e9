#!/bin/bash
# kernel times of the three training kernels alone, f16x3 against f16x3lo8 (same box, alternating)
for i in 1 2; do
  BWD=f16x3 python tools/bwd_time.py
  BWD=f16x3lo8 python tools/bwd_time.py
done
