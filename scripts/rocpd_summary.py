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
    # avg over all launches (cold first launches included) + min and median, which show the steady state
    print("kernel,calls,total_ms,avg_ms,percent,min_ms,median_ms")
    durs = {}
    for name, d in db.execute("select name,end-start from kernels"):
        durs.setdefault(name, []).append(d)
    tot = sum(sum(v) for v in durs.values())
    for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
        print(f'"{name[:100]}",{len(v)},{sum(v) / 1e6:.4f},{sum(v) / len(v) / 1e6:.5f},{100 * sum(v) / tot:.2f},'
              f'{v[0] / 1e6:.5f},{med / 1e6:.5f}')
else:
    subs = sys.argv[3:]
    print("kernel,dispatch,counter,value")
    q = "select name,dispatch_id,counter_name,sum(counter_value) from pmc_events group by name,dispatch_id,counter_name order by dispatch_id"
    for name, disp, cname, val in db.execute(q):
        if not subs or any(s in name for s in subs):
            print(f'"{name[:60]}",{disp},{cname},{val}')
