#!/bin/bash
for o in "big_iters=4096" "big_iters=2048" "big_iters=1024" "big_iters=512" "big_iters=1024,chunk=32768,arena_gb=200"; do
  echo "=== $o"; PROBE_OPTS="$o" PROBE_TOP=0 timeout 120 python tools/probe_heavy.py 2>&1 | grep "C3 mix.*split=0"
done 2>&1 | tee gpurun_out/opts2.log
