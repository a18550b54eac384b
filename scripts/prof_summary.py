"""Summarise rocprofv3 rocpd sqlite outputs (kernel stats + PMC counters per kernel) as text."""
import sqlite3
import sys


def short(n):
    n = n.replace("ilqr::", "")
    return n if len(n) < 90 else n[:87] + "..."


def main(paths):
    for f in paths:
        con = sqlite3.connect(f)
        cur = con.cursor()
        print("== %s" % f)
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print("%-92s %6s %12s %12s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
        for n, c, s, a, mn, mx in rows:
            print("%-92s %6d %12.1f %12.2f %10.2f %10.2f %6.2f" % (short(n), c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
        crow = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection group by kernel_name, counter_name order by avg(value) desc").fetchall()
        if crow:
            print("%-92s %-18s %6s %16s %16s %16s" % ("kernel", "counter", "n", "avg", "min", "max"))
            for k, cn, c, a, mn, mx in crow:
                print("%-92s %-18s %6d %16.1f %16.1f %16.1f" % (short(k), cn, c, a, mn, mx))
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
