#!/usr/bin/env python3
"""GPU idle time between consecutive kernels of the adapter step, from a rocprofv3 --kernel-trace rocpd database."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
rows = [r for r in rows if r[0].startswith(("void k_", "k_"))]
gaps, busy = [], 0
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    busy += e0 - s0
    g = s1 - e0
    if 0 <= g < 200_000:          # ignore the pauses between bench phases
        gaps.append(g)
gaps.sort()
print(f"{len(rows)} adapter kernels, busy {busy/1e6:.2f} ms, {len(gaps)} gaps: total {sum(gaps)/1e6:.2f} ms, "
      f"median {gaps[len(gaps)//2]/1e3:.2f} us, p90 {gaps[int(len(gaps)*0.9)]/1e3:.2f} us, mean {sum(gaps)/len(gaps)/1e3:.2f} us")
print(f"idle share of (busy + gaps): {100*sum(gaps)/(busy+sum(gaps)):.1f} %")
