#!/bin/bash
# GPU box session: parity tests, kernel probe, bench
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
PROBE_KSTATS=1 timeout 400 python tools/probe_kernel.py > gpurun_out/probe.log 2>&1
echo "probe rc=$?" >> gpurun_out/probe.log
timeout 500 python bench.py > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/bench.log
