#!/bin/bash
# round 3: where emit_kernel's time goes - the library built with -DMIBN_EMIT_PROF (phase timers, see MIBN_TICK in csrc/emit_core.h):
#   hipcc ... -c -o /tmp/planner_host.o sorobn_amd/csrc/planner.cpp
#   hipcc ... -shared -DMIBN_EMIT_PROF -o sorobn_amd/libmibn_prof.so /tmp/planner_host.o sorobn_amd/csrc/engine.hip -lpthread -ldl
# alone (blocking calls) and beside the level kernels (the bench's two calls in flight)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for mode in "--sync" ""; do
  echo "== bench.py $mode"
  MIBN_LIB=$PWD/sorobn_amd/libmibn_prof.so python bench.py --steps 3 --warmup 2 $mode --no-cpu --no-configs --no-adaptive --threads 2 --opt gpu_emit=1 --opt emit_share=1 --opt trace=1 2>&1 | grep "emit phases\|mibn plan. chunk" | tail -12
done
