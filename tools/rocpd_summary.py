#!/usr/bin/env python3
"""Condense rocprofv3 (ROCm 7.2, rocpd sqlite output) results into small text summaries that can
be committed under profiles/.

    python tools/rocpd_summary.py stats  <bench_results.db>  > profiles/rNN_kernel_stats.txt
    python tools/rocpd_summary.py pmc    <bench_results.db>  > profiles/rNN_pmc_<counter>.txt

`stats`: per kernel (== rocprofv3 --stats 'top_kernels') and per (kernel, grid) shape.
`pmc`  : per (kernel, grid, counter): number of dispatches, mean/min/max counter value.
"""
import re
import sqlite3
import sys


def short(name: str, n: int = 70) -> str:
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    name = name.replace("unsigned short", "bf16").replace("long long", "i64")
    return name if len(name) <= n else name[: n - 3] + "..."


def ours(k: str) -> bool:
    """Kernels of this library (k_*, fl::*) and the hipBLASLt GEMMs they are compared with."""
    return k.startswith("k_") or k.startswith("fl::") or k.startswith("Custom_Cijk") or k.startswith("Cijk_")


def modes(v):
    """One kernel symbol launched on two problem shapes with the SAME grid (k_t1 at K = 1024 and K = 4736: the grid only depends
    on M) gives a bimodal counter: split at the geometric midpoint instead of reporting a meaningless mean."""
    lo, hi = min(v), max(v)
    if lo <= 0 or hi < 1.5 * lo:
        return [("", v)]
    mid = (lo * hi) ** 0.5
    return [(" [low mode]", [x for x in v if x < mid]), (" [high mode]", [x for x in v if x >= mid])]


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, grid_x, grid_y, workgroup_x, duration, vgpr_count, lds_size from kernels").fetchall()
    tot = sum(r[4] for r in rows)
    by_k, by_s = {}, {}
    for name, gx, gy, wx, dur, vg, lds in rows:
        k = short(name)
        by_k.setdefault(k, []).append(dur)
        by_s.setdefault((k, gx // max(wx, 1), gy, vg, lds), []).append(dur)
    print("# rocprofv3 --kernel-trace --stats (rocpd) -- durations in microseconds")
    print(f"# total kernel time {tot / 1e3:.1f} us over {len(rows)} dispatches\n")
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, d in sorted(by_k.items(), key=lambda kv: -sum(kv[1]))[:25]:
        print(f"{k:72s} {len(d):6d} {sum(d) / 1e3:12.1f} {sum(d) / len(d) / 1e3:10.2f} {min(d) / 1e3:10.2f} {max(d) / 1e3:10.2f} {100 * sum(d) / tot:6.2f}")
    print("\n# per (kernel, workgroups_x, grid_y) shape")
    print(f"{'kernel':60s} {'wg_x':>6s} {'grid_y':>6s} {'vgpr':>5s} {'lds':>6s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
    for (k, gx, gy, vg, lds), d in sorted(by_s.items(), key=lambda kv: -sum(kv[1]))[:40]:
        if not ours(k):
            continue
        print(f"{k:60s} {gx:6d} {gy:6d} {vg:5d} {lds:6d} {len(d):6d} {sum(d) / len(d) / 1e3:10.2f} {min(d) / 1e3:10.2f} {max(d) / 1e3:10.2f}")


def stats_all(db, top=45, after=None):
    """Every kernel of a whole-model step (library kernels included): top `top` by total time.  With `after`
    (a substring of a kernel name, e.g. naive_conv: MIOpen's find pass runs reference convolutions on the first call
    of each configuration) only dispatches that start after the LAST dispatch of a matching kernel are counted --
    the steady-state steps."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, duration, start from kernels").fetchall()
    cut = None
    if after:
        hits = [st for name, _, st in rows if after in name]
        cut = max(hits) if hits else None
        if cut is not None:
            rows = [r for r in rows if r[2] > cut]
    tot = sum(r[1] for r in rows)
    by_k = {}
    for name, dur, _ in rows:
        by_k.setdefault(short(name, 110), []).append(dur)
    span = (max(r[2] for r in rows) - min(r[2] for r in rows)) / 1e3 if rows else 0.0
    print("# rocprofv3 --kernel-trace --stats (rocpd) -- durations in microseconds")
    if cut is not None:
        print(f"# steady state only: dispatches after the last '{after}' kernel (MIOpen find pass of the first step)")
    print(f"# total kernel time {tot / 1e3:.1f} us over {len(rows)} dispatches, {len(by_k)} distinct kernels, "
          f"wall span {span:.1f} us\n")
    print(f"{'kernel':112s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for k, d in sorted(by_k.items(), key=lambda kv: -sum(kv[1]))[:int(top)]:
        print(f"{k:112s} {len(d):6d} {sum(d) / 1e3:12.1f} {sum(d) / len(d) / 1e3:10.2f} {100 * sum(d) / tot:6.2f}")


def gaps(db, delim="k_pack", min_us=20, top=30):
    """Where the GPU waits for the host in a whole-model step.  The trace is cut into steps at every dispatch of
    `delim` (one per optimizer step: the batched operand repack); only the steps within 5 % of the shortest one are
    analysed (the timed, unsynchronised steps -- warm-up, phase timing and profiling steps of bench.py are longer).
    Idle = intervals with nothing running (end of everything dispatched so far -> next start), attributed to the
    kernel that ENDS the wait."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if short(r[0]).startswith(delim)]
    steps = [(a, b) for a, b in zip(marks, marks[1:]) if b - a > 100]
    durs = [rows[b][1] - rows[a][1] for a, b in steps]
    keep = [sb for sb, d in zip(steps, durs) if d <= 1.05 * min(durs)]
    by_next, big, idle_total, busy_total, ctx_rows = {}, [], 0, 0, []
    for a, b in keep:
        busy_end = rows[a][2]
        for (n0, s0, e0), (n1, s1, e1) in zip(rows[a:b], rows[a + 1:b + 1]):
            busy_end = max(busy_end, e0)
            g = s1 - busy_end
            if g > 0:
                idle_total += g
                if g >= float(min_us) * 1e3:
                    k = short(n1, 80)
                    acc = by_next.setdefault(k, [0, 0])
                    acc[0] += g; acc[1] += 1
                    big.append((g, short(n0, 60), k))
                    ctx_rows.append((g, s1))
    n = max(len(keep), 1)
    span = sum(rows[b][1] - rows[a][1] for a, b in keep)
    print(f"# {len(steps)} steps between '{delim}' dispatches, {len(keep)} steady ones analysed "
          f"({span / n / 1e6:.2f} ms each): GPU idle {idle_total / n / 1e6:.2f} ms per step ({100 * idle_total / max(span, 1):.1f} %)")
    print(f"# idle intervals >= {min_us} us by the kernel that ends them (ms per step, count per step):")
    for k, (t, c) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:int(top)]:
        print(f"{t / n / 1e6:9.3f} {c / n:7.1f}  {k}")
    print("# the 20 longest waits (ms): previous kernel -> next kernel")
    for g, a_, b_ in sorted(big, reverse=True)[:20]:
        print(f"{g / 1e6:9.3f}  {a_}  ->  {b_}")
    print("# context of the 6 longest waits: 4 kernels before | 6 kernels after")
    starts = [r[1] for r in rows]
    import bisect
    for g, s1 in sorted(ctx_rows, reverse=True)[:6]:
        i = bisect.bisect_left(starts, s1)
        print(f"{g / 1e6:9.3f} ms")
        for r in rows[max(i - 4, 0):i]:
            print(f"             before  {short(r[0], 100)}")
        for r in rows[i:i + 6]:
            print(f"             after   {short(r[0], 100)}")


def seq(db, last=44):
    """The last `last` dispatches in start order: start (us from the first of them), duration, and the idle gap since the
    end of everything dispatched before -- what a dependent launch on one stream costs beyond its kernel time."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()[-int(last):]
    t0, busy_end, gap_sum, dur_sum = rows[0][1], rows[0][1], 0, 0
    print(f"{'kernel':60s} {'start_us':>10s} {'dur_us':>9s} {'gap_us':>8s}")
    for n, s0, e0 in rows:
        g = s0 - busy_end
        gap_sum += max(g, 0); dur_sum += e0 - s0
        print(f"{short(n, 60):60s} {(s0 - t0) / 1e3:10.2f} {(e0 - s0) / 1e3:9.2f} {g / 1e3:8.2f}")
        busy_end = max(busy_end, e0)
    print(f"# span {(busy_end - t0) / 1e3:.1f} us: kernels {dur_sum / 1e3:.1f} us, gaps {gap_sum / 1e3:.1f} us over {len(rows)} dispatches")


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, grid_size_x, grid_size_y, workgroup_size_x, counter_name, value "
                       "from counters_collection").fetchall()
    agg = {}
    for name, gx, gy, wx, cn, v in rows:
        k = short(name)
        if not ours(k):
            continue
        agg.setdefault((k, gx // max(wx, 1), gy, cn), []).append(v)
    print("# rocprofv3 --pmc (rocpd) -- per (kernel, workgroups_x, grid_y, counter); values as reported")
    print("# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x")
    print("# (MI355X_MICROARCH.md, HBM section) -- bench.py doubles it before comparing with byte counts.\n")
    print(f"{'kernel':60s} {'wg_x':>6s} {'grid_y':>6s} {'counter':>14s} {'n':>5s} {'mean':>14s} {'min':>14s} {'max':>14s}")
    for (k, gx, gy, cn), vall in sorted(agg.items()):
        for tag, v in modes(vall):
            print(f"{(k[:48] + tag):60s} {gx:6d} {gy:6d} {cn:>14s} {len(v):5d} {sum(v) / len(v):14.1f} {min(v):14.1f} {max(v):14.1f}")


def traffic(db_fetch, db_write):
    """JSON rows: HBM bytes per launch = 2 * FETCH_SIZE KiB * 1024 (gfx950 under-reports wide coalesced
    reads by exactly 2x, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB * 1024 (calibrated: k_t2 writes
    exactly M*N*2 bytes and WRITE_SIZE reports exactly that)."""
    import json
    vals = {}
    for db, cn in ((db_fetch, "FETCH_SIZE"), (db_write, "WRITE_SIZE")):
        cur = sqlite3.connect(db).cursor()
        for name, gx, gy, wx, v in cur.execute(
                "select kernel_name, grid_size_x, grid_size_y, workgroup_size_x, value from counters_collection "
                "where counter_name = ?", (cn,)):
            k = short(name)
            if ours(k):
                vals.setdefault((k, gx // max(wx, 1), gy), {}).setdefault(cn, []).append(v)
    out = []
    for (k, gx, gy), d in sorted(vals.items()):
        # the FETCH counter separates the shapes of a kernel launched with one grid on two problem sizes (k_t1: K = 1024 / 4736);
        # the two passes dispatch in the same order, so the write samples are split at the same positions
        fm = modes(d.get("FETCH_SIZE", [0]))
        fall, wall = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
        for tag, fv in fm:
            if len(fm) > 1 and len(wall) == len(fall):
                keep = set(i for i, x in enumerate(fall) if (x in fv))
                wv = [w for i, w in enumerate(wall) if i in keep] or [0]
            else:
                wv = wall
            f = sum(fv) / max(1, len(fv))
            w = sum(wv) / max(1, len(wv))
            out.append(dict(kernel=k + tag, wg_x=gx, grid_y=gy, fetch_kib_mean=round(f, 1), write_kib_mean=round(w, 1),
                            launches=len(fv), hbm_bytes=int(2 * f * 1024 + w * 1024)))
    print(json.dumps(out, indent=1))


def block_traffic(db_fetch, db_write, n_blocks):
    """HBM bytes of ONE block's forward + backward adapter calls: every dispatch of this library's kernels in the two PMC passes over
    tools/block_traffic_run.py (the operand repack k_pack excluded: it runs once per optimizer step for the whole model), summed and
    divided by the number of block passes the command ran.  Same corrections as `traffic`."""
    import json
    n_blocks = int(n_blocks)
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    by_k = {}
    for db, cn in ((db_fetch, "FETCH_SIZE"), (db_write, "WRITE_SIZE")):
        cur = sqlite3.connect(db).cursor()
        for name, v in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (cn,)):
            k = short(name)
            if not (k.startswith("k_") or k.startswith("fl::")) or k.startswith("k_pack"):
                continue
            tot[cn] += v
            by_k.setdefault(k[:60], {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})[cn] += v
    per = lambda d: int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / n_blocks)
    print(json.dumps({"hbm_bytes_per_block": per(tot), "fetch_kib_per_block": round(tot["FETCH_SIZE"] / n_blocks, 1),
                      "write_kib_per_block": round(tot["WRITE_SIZE"] / n_blocks, 1), "block_passes": n_blocks,
                      "by_kernel_bytes_per_block": {k: per(d) for k, d in sorted(by_k.items(), key=lambda kv: -per(kv[1]))},
                      "what": "fwd + bwd of one ViT block's two adapters (plain entry points, M = 41,472, r = 16): 2 x FETCH_SIZE + WRITE_SIZE "
                              "(KiB, separate rocprofv3 --pmc passes over tools/block_traffic_run.py), k_pack excluded"}, indent=1))


if __name__ == "__main__":
    {"stats": stats, "stats_all": stats_all, "gaps": gaps, "seq": seq, "pmc": pmc, "traffic": traffic, "block_traffic": block_traffic}[sys.argv[1]](*sys.argv[2:])
