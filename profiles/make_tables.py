#!/usr/bin/env python3
"""Refresh the tables of profiles/<tag>_suite.md and <tag>_linear.md from the JSON files bench_suite.py / bench_linear.py
wrote (copied to profiles/<tag>_suite.json / <tag>_linear.json); the prose around the tables is kept.
usage: python profiles/make_tables.py r01"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def swap_table(path, header_prefix, rows):
    text = open(path).read()
    lines = text.split("\n")
    i = next(k for k, l in enumerate(lines) if l.startswith(header_prefix))
    j = i + 2
    while j < len(lines) and lines[j].startswith("|"):
        j += 1
    open(path, "w").write("\n".join(lines[:i + 2] + rows + lines[j:]))


suite = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_suite.json")))
swap_table(os.path.join(ROOT, "profiles", f"{tag}_suite.md"), "| case |",
           [f"| {r['case']} | {r['us_per_call']:.2f} | {r['elems_per_s'] / 1e9:.1f} | {r['algorithmic_bytes_per_elem']} | {r['achieved_GBps']:.0f} | "
            f"{100 * r['frac_of_8TBps']:.1f} |" for r in suite])
lin = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_linear.json")))
swap_table(os.path.join(ROOT, "profiles", f"{tag}_linear.md"), "| layer |",
           [f"| {r['layer']} | {r['tokens']} | {r['in_features']} x {r['out_features']} | {r['f_linear_us']:.1f} | {r['bfplinear_us']:.1f} | "
            f"{r['bfplinear_cached_us']:.1f} | {r['weight_quant_us']:.1f} | {r['act_quant_us']:.1f} | "
            + (f"{r['packed_decode_us']:.1f}" if r.get('packed_decode_us') else "-") + " |" for r in lin])
pf = os.path.join(ROOT, "profiles", f"{tag}_prefill.json")
if os.path.exists(pf):
    rows = json.load(open(pf))
    swap_table(os.path.join(ROOT, "profiles", f"{tag}_prefill.md"), "| layer |",
               [f"| {r['layer']} | {r['tokens']} | {r['in_features']} x {r['out_features']} | {r['f_linear_us']:.1f} ({r['f_linear_tflops']:.0f}) | "
                f"{r['bfplinear_cached_us']:.1f} | **{r['packed_prefill_us']:.1f}** | {r['act_image_us']:.1f} | {r['mx8_gemm_us']:.1f} ({r['mx8_gemm_tflops']:.0f}) | "
                f"{r['packed_prefill_us'] / r['f_linear_us']:.2f} |" for r in rows])
mf = os.path.join(ROOT, "profiles", f"{tag}_model.json")
if os.path.exists(mf):
    rows = json.load(open(mf))
    modes, toks = [], []
    for r in rows:
        if r['mode'] not in modes: modes.append(r['mode'])
        if r['tokens'] not in toks: toks.append(r['tokens'])
    get = {(r['mode'], r['tokens']): r['us_per_forward'] for r in rows}
    swap_table(os.path.join(ROOT, "profiles", f"{tag}_model.md"), "| linear layers |",
               [f"| {m} | " + " | ".join(f"{get[(m, t)]:.0f}" for t in toks) + " |" for m in modes])
print("tables refreshed")
