cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-configs --no-adaptive --sync --opt streams=1 --opt trace=1 2>&1 | grep "mibn launch" | grep sweep | tail -90
