"""profiles/<tag>_rocprofv3_summary.txt -> profiles/<tag>_pmc.json: HBM traffic per launch of every kernel of the exact path.

    python tools/make_pmc_json.py profiles/r02_u_rocprofv3_summary.txt

FETCH_SIZE / WRITE_SIZE come from separate rocprofv3 --pmc passes of the bench command (tools/gpu_profile.sh).  On gfx950
FETCH_SIZE tallies a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM section): doubled for ve_level_kernel and ve_mfma_kernel, whose
waves read 512 contiguous bytes per instruction.  The sweep kernels read runs of 64 bytes, which the counter tallies in
full - register-staged loads and LDS-DMA alike: calibrated on a known byte count (tools/gpu_r03_session.sh,
profiles/r03_*_pmc_calibration.log: FETCH_SIZE 8.582 GB for 8.590 GB read, WRITE_SIZE 8.608 / 8.590), so no correction there.  The algorithmic bytes per launch of the same command are taken from
the bench line at the end of the summary.  bench.py's `roofline.traffic` reads
the newest profiles/r*_pmc.json.
"""
import json
import re
import sys

path = sys.argv[1]
text = open(path).read()
fetch, write = {}, {}
for m in re.finditer(r"pmc (FETCH_SIZE|WRITE_SIZE): (\S+?)\(.*? launches (\d+) avg value ([0-9.]+) KB", text):
    name = m.group(2).split("::")[-1]
    (fetch if m.group(1) == "FETCH_SIZE" else write)[name] = (int(m.group(3)), float(m.group(4)))
line = [l for l in text.splitlines() if l.startswith('{"metric"')]
bench = json.loads(line[-1]) if line else {}
per = {}
for name in sorted(set(fetch) & set(write)):
    f, w = fetch[name][1], write[name][1]
    k = bench.get("kernels", {}).get(name)
    corr = 1.0 if name in ("ve_sweep_kernel", "ve_sweep_dma_kernel") else 2.0  # (see the session's *_pmc_calibration.log; the segment
    # kernel's scattered 8-byte reads are neither: its ratio is indicative only - 1.5 % of the bytes)
    # The counter passes see EVERY launch of the command (warm-up steps and the short first chunk of the pipeline included), the
    # bench line books the timed steps only: compare totals - traffic of all launches against the algorithmic bytes of all
    # steps (the steps are i.i.d. request batches of one size: algorithmic bytes per step x (steps + warm-up)) - and express
    # both per launch of the counter pass.
    steps_all = (bench.get("steps", 0) + bench.get("warmup", 0)) / max(1, bench.get("steps", 1))
    alg_per_launch = (k["alg_GB"] * 1e9 * steps_all / fetch[name][0]) if k else None
    per[name] = {"launches_under_the_counters": fetch[name][0], "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w,
                 "fetch_correction": corr, "traffic_bytes_per_launch": corr * f * 1024 + w * 1024,
                 "alg_bytes_per_launch_same_run": alg_per_launch}
# the concurrent launches of a level as one unit (bench.py's roofline names it when option overlap is on): the counters of its
# kernels summed (the PMC passes serialise the kernels, so every kernel's bytes are its own)
grp = [n for n in per if n in ("ve_level_kernel", "ve_sweep_dma_kernel", "ve_segment_kernel", "ve_mfma_kernel") and per[n]["alg_bytes_per_launch_same_run"]]
if len(grp) >= 2:
    tot_t = sum(per[n]["traffic_bytes_per_launch"] * per[n]["launches_under_the_counters"] for n in grp)
    tot_a = sum(per[n]["alg_bytes_per_launch_same_run"] * per[n]["launches_under_the_counters"] for n in grp)
    levels = max(per[n]["launches_under_the_counters"] for n in grp)
    per["level:ve_level_kernel||ve_mfma_kernel||ve_sweep_dma_kernel||ve_segment_kernel"] = {"launches_under_the_counters": levels, "fetch_correction": "per kernel, see its entries",
                                                   "traffic_bytes_per_launch": tot_t / levels, "alg_bytes_per_launch_same_run": tot_a / levels,
                                                   "members": grp}
out = {"source": path + " (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of the bench command)",
       "note": "gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled for ve_level_kernel; ve_sweep_kernel's 64-byte runs are "
               "counted in full (calibration: profiles/r02_u_pmc_calibration.log); WRITE_SIZE as reported", "per_kernel": per}
dst = path.replace("_rocprofv3_summary.txt", "_pmc.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst)
for n, d in per.items():
    a = d["alg_bytes_per_launch_same_run"]
    print("  %-20s traffic %.3f GB / launch, algorithmic %.3f GB -> %.3f" % (n, d["traffic_bytes_per_launch"] / 1e9, (a or 0) / 1e9, d["traffic_bytes_per_launch"] / a if a else float("nan")))
