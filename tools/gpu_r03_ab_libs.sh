#!/bin/bash
# round 3: A/B of library variants (MIBN_LIB) - the register budgets of the device planner's kernels
#   hipcc ... -DMIBN_ORDER_WAVES_PER_EU=<w> -DMIBN_EMIT_WAVES_PER_EU=<w> -o sorobn_amd/libmibn_o<w>e<w>.so sorobn_amd/csrc/planner.cpp sorobn_amd/csrc/engine.hip
# usage: gpu_r03_ab_libs.sh "<lib>[:<bench args>];..."
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
IFS=';' read -ra SETS <<< "$1"
for rep in 1 2; do
for set in "${SETS[@]}"; do
  lib=${set%%:*}; args=""; [[ "$set" == *:* ]] && args=${set#*:}
  MIBN_LIB=$PWD/sorobn_amd/$lib timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu --no-configs --no-adaptive --threads 2 --opt gpu_emit=1 --opt emit_share=1 $args 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); b = d['breakdown_ms_per_step']
        print('%-20s %-44s %.0f q/s  ms/step %.1f  kernel %.1f plan %.1f' % ('$lib', '$args', d['value'], d['ms_per_step'], b['kernel_ms'], b['plan_ms']))
    elif 'rror' in l: print(l.rstrip()[:200])
"
done
done
