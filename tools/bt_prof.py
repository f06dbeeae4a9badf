"""Median duration of the backtrace kernels in a rocprofv3 database of tools/bt_ab.py (first half of the launches: random weights, second
half: blank-dominated):   rocprofv3 --kernel-trace -d DIR -o NAME -- python3 tools/bt_ab.py REF.so;  python tools/bt_prof.py DIR/NAME_results.db"""
import sqlite3
import statistics as st
import sys

c = sqlite3.connect(sys.argv[1])
names = [r[0] for r in c.execute("select distinct name from kernels where name like '%backtrace%'")]
for nm in names:
    d = [r[0] / 1e3 for r in c.execute("select end-start from kernels where name = ? order by start", (nm,))]
    h = len(d) // 2
    print("%-60s random %.1f us   blank-dominated %.1f us   (%d launches)" % (nm[:60], st.median(d[:h]), st.median(d[h:]), len(d)))
