"""Summarise rocprofv3 rocpd (.db) outputs into a small text report for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_trace/trace_results.db [fetch.db] [write.db]
"""
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6 "
                            "from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1.0
    out = ["kernel-trace --stats  (durations in ms)", f"{'kernel':60s} {'calls':>6s} {'total':>10s} {'avg':>9s} {'min':>9s} {'max':>9s} {'%':>6s}"]
    for n, c, t, a, mn, mx in rows:
        out.append(f"{n[:60]:60s} {c:6d} {t:10.3f} {a:9.3f} {mn:9.3f} {mx:9.3f} {100*t/total:6.2f}")
    sym = list(cur.execute("select kernel_name, arch_vgpr_count, accum_vgpr_count, sgpr_count, group_segment_size, private_segment_size from kernel_symbols"))
    for s in sym:
        if "mibn" in s[0]:
            out.append(f"  {s[0][:70]}: vgpr {s[1]} agpr {s[2]} sgpr {s[3]} lds {s[4]} scratch {s[5]}")
    return out


def level_groups(db):
    """The launches of a level run concurrently on up to three streams (option overlap): their unit of time is the level - from the
    earliest start to the latest end of its launches.  A launch of level L + 1 starts only after every launch of level L has ended,
    so the levels are the connected components of the union of the VE kernels' intervals."""
    cur = sqlite3.connect(db).cursor()
    iv = sorted((s, e) for n, s, e in cur.execute("select name, start, end from kernels") if any(k in n for k in ("ve_level_kernel", "ve_sweep", "ve_segment_kernel", "ve_mfma_kernel")))
    if not iv:
        return []
    comps, (cs, ce), busy = [], iv[0], 0
    for s, e in iv[1:]:
        if s <= ce:
            ce = max(ce, e)
        else:
            comps.append((cs, ce))
            cs, ce = s, e
    comps.append((cs, ce))
    tot = sum(e - s for s, e in comps) / 1e6
    single = sum(e - s for s, e in iv) / 1e6
    # the idle time between consecutive levels (no VE kernel running): gaps of less than 0.5 ms are level boundaries inside a chunk
    # or between the chunks of a pipelined stream - longer ones are the ends of calls / steps
    gaps = sorted((comps[i + 1][0] - comps[i][1]) / 1e3 for i in range(len(comps) - 1))
    short = [g for g in gaps if g < 500.0]
    extra = []
    if short:
        extra = [f"idle gaps between consecutive levels (< 0.5 ms: {len(short)} of {len(gaps)}): total {sum(short) / 1e3:.3f} ms = {100 * sum(short) / 1e3 / tot:.2f} % of the "
                 f"VE busy time, median {short[len(short) // 2]:.1f} us, mean {sum(short) / len(short):.1f} us, 90th percentile {short[int(0.9 * len(short))]:.1f} us"]
    return extra + [f"levels (connected components of the VE kernels' intervals = level:ve_level_kernel||ve_mfma_kernel||ve_sweep_dma_kernel||ve_segment_kernel): {len(comps)} "
            f"groups, total {tot:.3f} ms = GPU busy time of the VE kernels, avg {tot / len(comps):.3f} ms per level; the sum of the individual "
            f"kernel durations is {single:.3f} ms ({single / tot:.2f} x: concurrent launches share the chip)"]


def pmc_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, counter_name, count(*), avg(counter_value), avg(duration)/1e6 from pmc_events "
                            "group by name, counter_name order by 4 desc"))
    out = []
    for n, c, k, v, d in rows:
        if "mibn" in n:
            out.append(f"pmc {c}: {n[:50]} launches {k} avg value {v:.1f} KB = {v*1024/1e9:.2f} GB per launch (avg kernel {d:.2f} ms under the counter pass)")
    return out


if __name__ == "__main__":
    print("\n".join(kernel_stats(sys.argv[1])))
    print("\n".join(level_groups(sys.argv[1])))
    for db in sys.argv[2:]:
        print("\n".join(pmc_stats(db)))
