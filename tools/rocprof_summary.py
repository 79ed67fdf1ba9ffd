#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) result DB into the text summaries kept under profiles/:
per-kernel stats (count, total, avg, min, max -- what `--stats` prints) and the launch timeline of
the last step.   usage: rocprof_summary.py <results.db> <out.txt> [steps_in_run | auto]
(auto: the number of k_insert_points launches = forwards in the run, warm-up, probes and traced steps included)"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    steps = None
    if len(sys.argv) > 3:
        steps = sum(r[1] for r in rows if "k_insert_points" in r[0]) if sys.argv[3] == "auto" else int(sys.argv[3])
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary of {db.split('/')[-1]}\n")
        f.write(f"# total kernel time {tot:.1f} us over {sum(r[1] for r in rows)} dispatches"
                + (f"; {steps} forwards -> {tot / steps:.1f} us of kernels per forward\n" if steps else "\n"))
        f.write(f"{'kernel':<96} {'calls':>6} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}\n")
        for r in rows:
            f.write(f"{r[0][:96]:<96} {r[1]:>6} {r[2]:>11.1f} {r[3]:>9.2f} {r[4]:>9.2f} {r[5]:>9.2f} {100 * r[2] / tot:>6.2f}\n")
        tl = cur.execute("select name, start, (end-start)/1e3, grid_x/workgroup_x, grid_y, grid_z, vgpr_count, "
                         "accum_vgpr_count, lds_size from kernels order by start").fetchall()
        marks = [i for i, r in enumerate(tl) if "k_insert_points" in r[0]]
        if marks:
            i0 = marks[-1]
            t0 = tl[i0][1]
            f.write("\n# timeline of the last step (all kernels, us since its first kernel)\n")
            for r in tl[i0:]:
                f.write(f"{(r[1] - t0) / 1e3:>10.1f} {r[0].split('(')[0][-60:]:<60} {r[2]:>9.2f} us  "
                        f"grid=({r[3]},{r[4]},{r[5]}) vgpr={r[6]}+{r[7]} lds={r[8]}\n")


if __name__ == "__main__":
    main()
