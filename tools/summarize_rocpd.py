#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run stored in rocpd (sqlite) format.
  tools/summarize_rocpd.py <results.db> <out.md> "<title / command line>" """
import re
import sqlite3
import sys


def main():
    db, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, s.arch_vgpr_count, s.sgpr_count "
         f"from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nSource: `rocprofv3 --kernel-trace --stats` (rocpd database summarised by tools/summarize_rocpd.py). "
                f"Total GPU kernel time {tot:.1f} ms.\n\n| kernel | calls | total ms | avg us | % | VGPRs |\n|---|---|---|---|---|---|\n")
        for n, c, t, a, vg, sg in rows[:30]:
            n = re.sub(r"^void ", "", n)
            f.write(f"| `{n[:110]}` | {c} | {t:.2f} | {a:.1f} | {100 * t / tot:.1f} | {vg} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
