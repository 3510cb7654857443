#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes
DIR/NAME_results.db on ROCm 7.2) into a small text table that can be committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_e2e/e2e_results.db > profiles/r01_e2e_kernel_stats.txt
"""
import sqlite3
import sys

import numpy as np


def main(path, like="%"):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count),"
        " max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x)"
        " from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_ns':>9} {'min_ns':>8} {'max_ns':>9} {'pct':>6} {'vgpr':>5} {'agpr':>5}"
          f" {'sgpr':>5} {'lds':>6} {'scr':>4} {'grid':>8} {'wg':>4}  name")
    for r in [r for k, r in enumerate(rows) if k < 12 or r[0].startswith(("qr::", "void qr::", "_ZN2qr"))]:   # top 12 + every kernel of this library
        print(f"{r[1]:7d} {r[2]/1e6:10.3f} {r[3]:9.0f} {r[4]:8d} {r[5]:9d} {100*r[2]/total:6.2f} {r[6]:5d} {r[7]:5d}"
              f" {r[8]:5d} {r[9]:6d} {r[10]:4d} {r[11]:8d} {r[12]:4d}  {r[0][:110]}")
    k = np.array(cur.execute("select start, end from kernels where name like '%step_kernel%' or name like '%rollout_kernel%'"
                             " order by start").fetchall())
    if len(k) > 10:
        d = k[:, 1] - k[:, 0]
        gaps = k[1:, 0] - k[:-1, 1]
        print(f"# env step kernels: n={len(k)} duration median {np.median(d):.0f} ns, p10 {np.percentile(d,10):.0f},"
              f" p90 {np.percentile(d,90):.0f}, mean {d.mean():.0f}; gap to next launch median {np.median(gaps):.0f} ns")


if __name__ == "__main__":
    main(*sys.argv[1:])
