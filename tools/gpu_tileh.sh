#!/bin/bash
for th in 4 8 16 32 64; do
  echo "=== tile_h=$th"
  PROBE_OPTS="tile_h=$th" PROBE_TOP=2 timeout 120 python tools/probe_heavy.py 2>&1 | grep -v "^$" | head -4
done 2>&1 | tee gpurun_out/tileh.log
