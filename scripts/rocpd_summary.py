#!/usr/bin/env python3
"""Summaries of rocprofv3 rocpd (sqlite) outputs.
  rocpd_summary.py stats  RESULTS.db             -> kernel stats CSV on stdout (top_kernels view)
  rocpd_summary.py stats_ms RESULTS.db           -> kernel stats in ms from the kernels view (start/end ns)
  rocpd_summary.py pmc    RESULTS.db [substr..]  -> per-dispatch counter sums CSV on stdout"""
import sqlite3, sys
mode, path = sys.argv[1], sys.argv[2]
db = sqlite3.connect(path)
if mode == "stats":
    print("kernel,calls,total_ms,avg_ms,percent")
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
        print(f'"{name[:100]}",{calls},{total / 1e6:.4f},{avg / 1e6:.5f},{pct:.2f}')
elif mode == "stats_ms":
    print("kernel,calls,total_ms,avg_ms,percent")
    rows = list(db.execute("select name,count(*),sum(end-start),avg(end-start) from kernels group by name order by sum(end-start) desc"))
    tot = sum(r[2] for r in rows)
    for name, calls, total, avg in rows:
        print(f'"{name[:100]}",{calls},{total / 1e6:.4f},{avg / 1e6:.5f},{100 * total / tot:.2f}')
else:
    subs = sys.argv[3:]
    print("kernel,dispatch,counter,value")
    q = "select name,dispatch_id,counter_name,sum(counter_value) from pmc_events group by name,dispatch_id,counter_name order by dispatch_id"
    for name, disp, cname, val in db.execute(q):
        if not subs or any(s in name for s in subs):
            print(f'"{name[:60]}",{disp},{cname},{val}')
