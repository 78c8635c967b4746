#!/usr/bin/env python3
"""GPU idle time between kernels in a rocprofv3 --kernel-trace run (rocpd SQLite database).

    python tools/gap_report.py gpurun_out/c14/prof_train/r_results.db [min_gap_us]

Prints busy / idle totals of the traced span and the largest gaps with the kernels either side.
"""
import sqlite3
import sys


def main(path, min_gap_us=5.0):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
    s_col = "start" if "start" in cols else [c for c in cols if "start" in c][0]
    e_col = "end" if "end" in cols else [c for c in cols if "end" in c][0]
    rows = list(cur.execute(f'select name, "{s_col}", "{e_col}" from kernels order by "{s_col}"'))
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    gaps = []
    last_end, last_name = rows[0][2], rows[0][0]
    for name, s, e in rows[1:]:
        if s > last_end:
            gaps.append((s - last_end, last_name, name))
        if e > last_end:
            last_end, last_name = e, name
    idle = sum(g for g, _, _ in gaps)
    print(f"span {span/1e6:.2f} ms, kernel busy {busy/1e6:.2f} ms, idle between kernels {idle/1e6:.2f} ms over {len(gaps)} gaps")
    small = sum(g for g, _, _ in gaps if g < min_gap_us * 1e3)
    print(f"gaps below {min_gap_us} us: {small/1e6:.2f} ms total")
    hist = {}
    for g, a, b in gaps:
        if g >= min_gap_us * 1e3:
            k = (a[:60], b[:60])
            n, t = hist.get(k, (0, 0))
            hist[k] = (n + 1, t + g)
    for (a, b), (n, t) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{t/1e3:9.1f} us in {n:3d} gaps   after {a}  ->  before {b}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 5.0)
