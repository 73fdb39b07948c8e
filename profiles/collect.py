#!/usr/bin/env python3
"""Turn a gpurun_out/prof/ directory (written by tools_dev/prof.sh on the GPU box) into the committed
summaries under profiles/: kernel stats CSV, PMC traffic markdown and traffic.json (read by bench.py).
usage: python profiles/collect.py <round-tag> [bench-json]"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prof = os.path.join(ROOT, "gpurun_out", "prof")
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)      # gpurun merges runs into the same directory: take the latest
stats = newest(os.path.join(prof, "trace", "*", "*_kernel_stats.csv"))
shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_headline.csv"))
top = next(r for r in csv.DictReader(open(stats)) if "k_fused_flat" in r["Name"])
vals = {}
for kind in ("fetch", "write"):
    f = newest(os.path.join(prof, f"pmc_{kind}", "*", "*_counter_collection.csv"))
    rows = [r for r in csv.DictReader(open(f)) if "k_fused_flat" in r["Kernel_Name"]]
    v = [float(r["Counter_Value"]) for r in rows]
    vals[kind] = (sum(v) / len(v), len(v), rows[0])
fetch_kb, write_kb = vals["fetch"][0], vals["write"][0]
hbm = (2 * fetch_kb + write_kb) * 1024
r0 = vals["fetch"][2]
key = "4096x11008_bf16_m3_b64_2:4_s_dropin"
tj = {key: {"hbm_bytes_per_launch": hbm, "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb,
            "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on 16 B/lane streams -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; both in KB",
            "source": f"profiles/{tag}_pmc_headline.md (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)"}}
json.dump(tj, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
bench = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
bt = json.load(open(os.path.join(prof, "bench_trace.json")))
md = f"""# {tag} -- rocprofv3 evidence for the headline kernel

Commands (tools_dev/prof.sh; each counter in its own pass, kernel-trace/stats in a third; run from /tmp on the GPU box):

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 --eager
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 --eager

Kernel: `{r0['Kernel_Name']}` (grid {r0['Grid_Size']} work-items, VGPR {r0['VGPR_Count']}, SGPR {r0['SGPR_Count']}, LDS {r0['LDS_Block_Size']} B)
Workload: LLaMA-7B down_proj [4096,11008] bf16 -> 2:4 + HBFP4 block 64, drop-in output (45,088,768 elements per launch).

## kernel-trace stats ({tag}_kernel_stats_headline.csv)

{top['Calls']} calls, average {float(top['AverageNs'])/1e3:.2f} us, min {float(top['MinNs'])/1e3:.2f} us, max {float(top['MaxNs'])/1e3:.2f} us
({float(top['Percentage']):.2f} % of GPU time in the run); bench.py's own HIP-event figure in the same profiled process: {bt['roofline']['avg_launch_us']:.2f} us per launch.
"""
if bench:
    md += f"Un-profiled bench.py on the same box: {bench['roofline']['avg_launch_us']:.2f} us per launch = {bench['roofline']['achieved']:.0f} GB/s = {bench['roofline']['frac']*100:.1f} % of 8 TB/s.\n"
md += f"""
## PMC traffic

| counter | dispatches | mean per launch (raw, KB) | bytes per launch |
|---|---|---|---|
| FETCH_SIZE | {vals['fetch'][1]} | {fetch_kb:.1f} | {2*fetch_kb*1024:,.0f} (x2 gfx950 correction) |
| WRITE_SIZE | {vals['write'][1]} | {write_kb:.1f} | {write_kb*1024:,.0f} |
| total HBM-side traffic | | | **{hbm:,.0f}** |
| algorithmic bytes (2 B in + 2 B out per element) | | | 180,355,072 |

Traffic / algorithmic = {hbm/180355072:.4f}: every input byte is read once and every output byte written once (the excess on
the read side is the per-workgroup load of the 729-byte N:M table and the 320-byte exponent table).
"""
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_headline.md"), "w").write(md)
print(md)
