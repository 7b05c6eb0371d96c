cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-configs --no-adaptive --opt trace=1 2>&1 | grep -E "mibn gap|metric" | cut -c1-400 | tail -30
