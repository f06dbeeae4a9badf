"""Median / minimum duration of the kernels whose name contains PATTERN in a rocprofv3 database:
    rocprofv3 --kernel-trace -d DIR -o NAME -- python3 <script>;  python tools/kprof.py DIR/NAME_results.db PATTERN [PATTERN ...]"""
import sqlite3
import statistics as st
import sys

c = sqlite3.connect(sys.argv[1])
for pat in sys.argv[2:]:
    for (nm,) in c.execute("select distinct name from kernels where name like ?", ("%" + pat + "%",)).fetchall():
        d = [r[0] / 1e3 for r in c.execute("select end-start from kernels where name = ? order by start", (nm,))]
        print("%-70s median %.1f us  min %.1f us  (%d launches)" % (nm[:70], st.median(d), min(d), len(d)))
