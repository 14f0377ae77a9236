"""Turn a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME -- CMD -> NAME_results.db) into the
small text summary that is committed under profiles/."""
import sqlite3
import sys


def main(db_path, title):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    print("# %s" % title)
    print()
    print("| kernel | calls | avg ms | total ms | % | grid | wg | VGPR | AGPR | SGPR | LDS B | scratch B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    res = {}
    for r in cur.execute("select name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size "
                         "from kernels group by name"):
        res[r[0]] = r[1:]
    for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if pct < 0.001:
            continue
        short = name.split("(")[0].replace("void ", "")
        if len(short) > 60:
            short = short[:57] + "..."
        g = res.get(name, ("?",) * 7)
        print("| `%s` | %d | %.3f | %.3f | %.2f | %s | %s | %s | %s | %s | %s | %s |" % (
            short, calls, avg / 1e3, total / 1e3, pct, g[0], g[1], g[2], g[3], g[4], g[5], g[6]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
