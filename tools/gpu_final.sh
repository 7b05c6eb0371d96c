#!/bin/bash
# GPU box session for the committed measurements of a round: tools/gpu_final.sh <tag>
TAG=${1:-r01_x}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out; mkdir -p $OUT
bash tools/gpu_profile.sh $TAG > $OUT/${TAG}_profile_session.log 2>&1
timeout 600 python bench.py > $OUT/${TAG}_bench.log 2>&1
for ne in 1 2 8 16; do
  echo "n_evidence=$ne" >> $OUT/${TAG}_n_evidence_variants.log
  timeout 300 python bench.py --no-cpu --n-evidence $ne 2>&1 | tail -1 | cut -c1-700 >> $OUT/${TAG}_n_evidence_variants.log
done
PROBE_SETS="split_kinds=0;split_kinds=1;chain=0,split_kinds=0;chain=0,split_kinds=1" timeout 300 python tools/probe_opts.py > $OUT/${TAG}_probe_classes.log 2>&1
timeout 600 python tools/bench_configs.py > $OUT/${TAG}_other_configs.log 2>&1
tail -1 $OUT/${TAG}_bench.log | cut -c1-600
cat $OUT/${TAG}_rocprofv3_summary.txt | cut -c1-200
