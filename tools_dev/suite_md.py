#!/usr/bin/env python3
"""profiles/<tag>_suite.md from bench_suite.py's JSON: usage suite_md.py <suite.json> <previous suite.json | -> <title> > out.md (the prose around the table is written by hand)"""
import json, sys
rows = json.load(open(sys.argv[1]))
old = {r['case']: r for r in json.load(open(sys.argv[2]))} if sys.argv[2] != '-' else {}
print(f"# {sys.argv[3]}\n")
print("| case | us | Gelem/s | B/elem | GB/s | % of 8 TB/s | same kernel, caller-owned rotating outputs: us (%) | previous table (us) |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    us = r.get('us_per_call', r.get('us_per_pass'))
    o = old.get(r['case'])
    ous = (o.get('us_per_call', o.get('us_per_pass')) if o else None)
    rot = r.get('us_per_call_rotating_outputs')
    rots = f"{rot:.2f} ({r['frac_of_8TBps_rotating_outputs'] * 100:.1f})" if rot else ""
    print(f"| {r['case']} | {us:.2f} | {r['elems_per_s'] / 1e9:.1f} | {r['algorithmic_bytes_per_elem']} | {r['achieved_GBps']:.0f} | {r['frac_of_8TBps'] * 100:.1f} | {rots} | {ous:.2f} |" if ous else
          f"| {r['case']} | {us:.2f} | {r['elems_per_s'] / 1e9:.1f} | {r['algorithmic_bytes_per_elem']} | {r['achieved_GBps']:.0f} | {r['frac_of_8TBps'] * 100:.1f} | {rots} | |")
