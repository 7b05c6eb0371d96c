#!/bin/bash
# Round-3 session a: access-pattern ceilings, the LDS-DMA sweep kernel in isolation and in the C3 mix, parity of the sweep tests.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/r03_a
mkdir -p $OUT
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)"; rocm-smi --showclocks 2>/dev/null | head -20) > $OUT/host.txt 2>&1
timeout 120 tools/ubench/pattern_copy 2048 > $OUT/pattern_copy.log 2>&1; tail -20 $OUT/pattern_copy.log
for it in 8 16; do timeout 200 tools/ubench/sweep_real 2048 $it >> $OUT/sweep_real.log 2>&1; done; cat $OUT/sweep_real.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep or grid10x10 or c3_stream or heavy or chain_form or wide_grids" > $OUT/pytest_sweep.log 2>&1; tail -8 $OUT/pytest_sweep.log
for args in "--opt sweep_dma=0" "--opt sweep_dma=1" "--opt sweep_dma=2" "--opt sweep_dma=1 --opt sweep_iters=16" "--opt sweep_dma=1 --opt sweep_adapt=2048"; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu --no-configs --no-adaptive $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; b = d['breakdown_ms_per_step']
        print('%-44s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f  GB/s(all) %.0f  MB/query %.2f' % ('$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms'], r['all_kernels_GBps'], r['alg_bytes_per_query'] / 1e6))
        for k, v in d['kernels'].items(): print('      %-20s launches %5d ms %8.1f  GB %8.1f  -> %6.0f GB/s' % (k, v['launches'], v['ms'], v['alg_GB'], v['GBps']))
    elif 'rror' in l: print(l.rstrip()[:300])
" | tee -a $OUT/bench_ab.log
done
