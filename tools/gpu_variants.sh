#!/bin/bash
# run tools/probe_heavy.py against every kernel-variant library sorobn_amd/libmibn_*.so
mkdir -p gpurun_out
for lib in sorobn_amd/libmibn_*.so; do
  echo "=== $lib"
  MIBN_LIB=$PWD/$lib PROBE_TOP=${PROBE_TOP:-3} timeout 120 python tools/probe_heavy.py 2>&1 | grep -v "^$"
done 2>&1 | tee gpurun_out/variants.log
