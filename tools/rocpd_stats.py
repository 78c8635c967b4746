#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace --stats run
whose output is a rocpd SQLite database; prints the table that goes under profiles/.

    python tools/rocpd_stats.py gpurun_out/prof/r1_results.db > profiles/r01_xyz.stats.txt
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                            "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name "
                            "order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats  ({path})")
    print(f"# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6} {'vgpr':>5} {'agpr':>5} {'lds':>7}  name")
    for name, n, tot, avg, mn, mx, vg, ag, lds in rows:
        print(f"{n:6d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*tot/total:6.2f} {vg or 0:5d} {ag or 0:5d} "
              f"{lds or 0:7d}  {name[:150]}")


if __name__ == "__main__":
    main(sys.argv[1])
