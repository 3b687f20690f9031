"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the plain-text per-kernel
summary kept under profiles/.   usage: python tools/rocpd_summary.py results.db [out.txt]"""
import sqlite3
import sys


def summarize(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
                      "max(grid_x), max(grid_y), max(workgroup_x) from kernels group by name "
                      "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["%-58s %6s %12s %12s %12s %12s %6s %5s %5s %5s %7s  %s" %
             ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct", "vgpr", "agpr", "sgpr", "lds", "grid/wg")]
    for r in rows:
        name = r[0].split("(")[0][-58:]
        lines.append("%-58s %6d %12.3f %12.4f %12.4f %12.4f %6.2f %5d %5d %5d %7d  %dx%d/%d" %
                     (name, r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, 100.0 * r[2] / total,
                      r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0, r[11] or 0, r[12] or 0))
    return "\n".join(lines)


if __name__ == "__main__":
    text = summarize(sys.argv[1])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)
