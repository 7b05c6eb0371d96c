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
    for db in sys.argv[2:]:
        print("\n".join(pmc_stats(db)))
