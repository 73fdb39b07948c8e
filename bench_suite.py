#!/usr/bin/env python3
"""bench_suite.py -- per-configuration timings of the hot path on one MI355X (all BASELINE.json configs).
Not the driver contract (that is bench.py); writes a JSON list to --out and a table to stdout.
Each case: L launches captured in one hipGraph over R rotating inputs, HIP-event time / L, median of rounds."""
import argparse, json, os, statistics, sys
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import torch
import quantization_sparsity_interplay_amd as pkg
from quantization_sparsity_interplay_amd.bfp import bfp_ops
from quantization_sparsity_interplay_amd import native

DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def cfg(**kw):
    base = dict(mant_bits=3, epsilon=1e-8, rounding_mode='determ', device='cuda', block_size=64, num_format='bfp',
                weight_mant_bits=15, in_sparsity=False, w_sparsity=False, grad_sparsity=False, sparsity_frac=0.5, N=2, M=4,
                sparsity_num_format='bfp', first='s', sparsity_mode='structured')
    base.update(kw)
    return base


CASES = [
    # name, rows, cols, dtype, bytes/elem (algorithmic), callable-maker
    ("cfg1 OPT-125m [768,768] f32 HBFP8 b32 dense", 768, 768, "f32", 8, dict(mant_bits=7, block_size=32)),
    ("cfg2 q_proj [4096,4096] bf16 HBFP4 b64 dense", 4096, 4096, "bf16", 4, dict()),
    ("cfg3 down_proj [4096,11008] bf16 HBFP4 b64 2:4 s (headline)", 4096, 11008, "bf16", 4, dict(w_sparsity=True)),
    ("cfg3 gate_proj [11008,4096] bf16 HBFP4 b64 2:4 s", 11008, 4096, "bf16", 4, dict(w_sparsity=True)),
    ("cfg3 down_proj bf16 2:4 q (quantize first)", 4096, 11008, "bf16", 4, dict(w_sparsity=True, first='q')),
    ("cfg3 down_proj f16 2:4 s", 4096, 11008, "f16", 4, dict(w_sparsity=True)),
    ("cfg3 down_proj f32 2:4 s", 4096, 11008, "f32", 8, dict(w_sparsity=True)),
    ("cfg3 down_proj bf16 HBFP8 b64 2:4 s", 4096, 11008, "bf16", 4, dict(w_sparsity=True, mant_bits=7)),
    ("cfg3 down_proj bf16 2:4 only (fp32 format)", 4096, 11008, "bf16", 4, dict(w_sparsity=True, sparsity_num_format='fp32')),
    ("cfg4 13B q_proj [5120,5120] bf16 HBFP4 + 50% unstructured s", 5120, 5120, "bf16", 4, dict(w_sparsity=True, sparsity_mode='unstructured')),
    ("cfg4 13B gate [13824,5120] bf16 HBFP4 + 50% unstructured s", 13824, 5120, "bf16", 4, dict(w_sparsity=True, sparsity_mode='unstructured')),
    ("cfg4 13B q_proj f32 HBFP4 + 50% unstructured s", 5120, 5120, "f32", 8, dict(w_sparsity=True, sparsity_mode='unstructured')),
    ("cfg4 13B q_proj [5120,5120] bf16 HBFP4 + 50% unstructured q (quantize first: three launches)", 5120, 5120, "bf16", 4, dict(w_sparsity=True, sparsity_mode='unstructured', first='q')),
    ("cfg5 ViT-L fc1 [4096,1024] f32 HBFP8 b16 1:4", 4096, 1024, "f32", 8, dict(mant_bits=7, block_size=16, N=1, M=4, w_sparsity=True)),
    ("cfg5 ViT-L act [8x197,1024] f32 HBFP8 b16 dense (identifier in)", 8 * 197, 1024, "f32", 8, dict(mant_bits=7, block_size=16, N=1, M=4, w_sparsity=True, _ident='in')),
    ("down_proj bf16 HBFP4 b64 4:8 s (N:8 in the flat kernel)", 4096, 11008, "bf16", 4, dict(w_sparsity=True, N=4, M=8)),
    ("down_proj bf16 HBFP4 b64 4:8 q, tie-heavy input (every group through the N:8 rank table)", 4096, 11008, "bf16", 4, dict(w_sparsity=True, N=4, M=8, first='q', _prequant=True)),
    ("down_proj f32 HBFP4 b64 4:8 s (N:8 in the flat kernel, DPP pair exchange)", 4096, 11008, "f32", 8, dict(w_sparsity=True, N=4, M=8)),
    ("general path: [4096,11000] bf16 HBFP4 b64 dense (ragged rows)", 4096, 11000, "bf16", 4, dict()),
    ("stochastic rounding: down_proj bf16 HBFP4 2:4 s (fp32 out, as the reference returns)", 4096, 11008, "bf16", 6, dict(w_sparsity=True, rounding_mode='stoc')),
    ("int8 per-row: down_proj bf16 weights (fp32 out)", 4096, 11008, "bf16", 6, dict(sparsity_num_format='int', mant_bits=8)),
    ("int8 per-column: activation [4096,4096] bf16 (fp32 out)", 4096, 4096, "bf16", 6, dict(sparsity_num_format='int', mant_bits=8, _ident='in')),
    ("cfg3 tie-heavy input (already HBFP4) bf16 2:4 q", 4096, 11008, "bf16", 4, dict(w_sparsity=True, first='q', _prequant=True)),
    # packed outputs (north_star "packed int4 stores"): 2 B read + 0.5 B codes + 1/64 B exponents; the matrix unit's image: 2 + 1 + 1/64
    ("cfg3 down_proj bf16 2:4 s, PACKED output (4-bit codes + int8 exponents)", 4096, 11008, "bf16", 2.516, dict(w_sparsity=True, _packed=4)),
    ("cfg2 q_proj bf16 dense, PACKED output (4-bit codes + int8 exponents)", 4096, 4096, "bf16", 2.516, dict(_packed=4)),
    ("activation [2048,11008] bf16 -> e4m3 + E8M0 image (operand of the block-scaled matrix unit)", 2048, 11008, "bf16", 3.016, dict(_packed=8)),
]


def model_rows(args, dev):
    """whole-configuration passes (BASELINE.json configs 1, 3, 4, 5 say "all Linear weights"): every Linear weight of the
    model through ONE call of float_to_bfp_blocked_many (a list of tensors per launch), timed with HIP events around the
    eager call; synthetic weights drawn on the device (randn * 0.02).  B/elem as in the single-tensor rows."""
    def llama(h, inter, layers):
        return [(h, h)] * 4 * layers + [(inter, h)] * 2 * layers + [(h, inter)] * layers
    vit = [(1024, 1024)] * 4 * 24 + [(4096, 1024)] * 24 + [(1024, 4096)] * 24
    opt = [(768, 768)] * 4 * 12 + [(3072, 768)] * 12 + [(768, 3072)] * 12
    models = [
        ("cfg1 OPT-125m: all 72 Linear weights, f32 HBFP8 b32 dense", opt, "f32", 8, dict(mant_bits=7, block_size=32), 'w'),
        ("cfg3 LLaMA-7B: all 224 Linear weights, bf16 HBFP4 b64 2:4 s", llama(4096, 11008, 32), "bf16", 4, dict(w_sparsity=True), 'w'),
        ("cfg4 LLaMA-13B: all 280 Linear weights, bf16 HBFP4 b64 + 50% unstructured s", llama(5120, 13824, 40), "bf16", 4,
         dict(w_sparsity=True, sparsity_mode='unstructured'), 'w'),
        ("cfg5 ViT-L/16: all 144 attention+MLP weights, f32 HBFP8 b16 1:4", vit, "f32", 8, dict(mant_bits=7, block_size=16, N=1, M=4, w_sparsity=True), 'w'),
        ("cfg5 ViT-L/16: the 144 Linear inputs of one forward [8,197,C], f32 HBFP8 b16 (identifier in)",
         [(8 * 197, 1024)] * 5 * 24 + [(8 * 197, 4096)] * 24, "f32", 8, dict(mant_bits=7, block_size=16, N=1, M=4, w_sparsity=True), 'in'),
    ]
    out = []
    for name, shapes, dname, bpe, kw, ident in models:
        if args.only and args.only not in name:
            continue
        dt = DT[dname]
        c = cfg(**kw)
        g = torch.Generator(device=dev).manual_seed(1234)
        ws = [(torch.randn(r, cc, generator=g, device=dev) * 0.02).to(dt) for r, cc in shapes]
        numel = sum(w.numel() for w in ws)

        prep = bfp_ops.PreparedMany(ws, identifier=ident, **c)      # outputs + descriptors bound once (the weights stay put)

        def run():
            return prep.run()
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(max(3, args.rounds)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); res = run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
            del res
        us = statistics.median(ts)
        gbps = numel * bpe / us / 1e3
        out.append(dict(case=name, tensors=len(ws), elems=numel, dtype=dname, us_per_pass=us, elems_per_s=numel / us * 1e6,
                        algorithmic_bytes_per_elem=bpe, achieved_GBps=gbps, frac_of_8TBps=gbps / 8000, launch="eager, prepared list (one launch per 64 tensors)"))
        print(f"{name:100s} {us:10.1f} us/pass  {numel/us/1e3:8.1f} Gelem/s  {gbps:7.0f} GB/s ({gbps/80:5.1f}%)", flush=True)
        del ws
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "suite.json"))
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--models", action="store_true", help="also (or with --models-only: only) the whole-model passes")
    ap.add_argument("--models-only", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no hipGraph (counter collection: every dispatch is its own record)")
    ap.add_argument("--no-list-graph", action="store_true", help="whole-model rows: the unstructured list issues its launches eagerly (kernel traces: the profiler serialises the nodes of a replayed graph)")
    ap.add_argument("--tune-grid", type=int, default=0, help="A/B: cap on workgroups of the streaming kernels (bfpq_tune key 0)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.no_list_graph:
        native.PruneQuantizeList.GRAPH_FROM = 1 << 30
    if args.tune_grid:
        assert native.load_library().bfpq_tune(0, args.tune_grid) == 0
    results = []
    if args.models_only:
        results = model_rows(args, dev)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(results, open(args.out, "w"), indent=1)
        return
    for name, rows, cols, dname, bpe, kw in CASES:
        if args.only and args.only not in name:
            continue
        kw = dict(kw)
        ident = kw.pop("_ident", "w")
        prequant = kw.pop("_prequant", False)
        packed = kw.pop("_packed", 0)
        c = cfg(**kw)
        dt = DT[dname]
        numel = rows * cols
        R = max(2, min(8, int(600e6 // (numel * (4 if dt == torch.float32 else 2))) or 2))
        ins = [(torch.randn(rows, cols, generator=torch.Generator().manual_seed(1234 + r)) * 0.02).to(dt).to(dev) for r in range(R)]
        if prequant:      # inputs already on the HBFP4 grid: ~38 % of the 2:4 groups tie at the boundary
            ins = [bfp_ops.float_to_bfp_blocked(x, **cfg(), identifier='w') for x in ins]
        L = args.launches

        def run():
            for i in range(L):
                if packed == 4:
                    sp = c['w_sparsity']
                    bfp_ops.float_to_bfp_packed(ins[i % R], c['mant_bits'], c['block_size'], c['epsilon'], c['N'] if sp else 0, c['M'] if sp else 0, c['first'], 4)
                elif packed == 8:
                    native.quantize_mx8(ins[i % R], c["mant_bits"], c["epsilon"])
                else:
                    bfp_ops.float_to_bfp_blocked(ins[i % R], **c, identifier=ident)
        run()
        torch.cuda.synchronize()
        mode = "hipGraph"
        try:
            if args.eager:
                raise RuntimeError("--eager")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            replay = g.replay
        except Exception as e:                                   # not capturable: time the eager path
            mode = f"eager ({type(e).__name__})"
            torch.cuda.synchronize()
            replay = run
        replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / L)
        us = statistics.median(ts)
        gbps = numel * bpe / us / 1e3
        # the same kernel through native.quantize_nm with R caller-owned OUTPUT buffers in rotation (what bench.py and the C-ABI harnesses do):
        # the reference-shaped call above allocates a fresh output per call, i.e. 40 distinct output tensors per graph replay
        us_rot = None
        structured = c['sparsity_mode'] == 'structured' and c['sparsity_num_format'] == 'bfp' and c['rounding_mode'] == 'determ' and packed in (0, 4) and not args.eager
        if structured:
            sp = c['w_sparsity'] and ident == 'w'
            N_, M_ = (c['N'], c['M']) if sp else (0, 0)
            outs = [torch.empty_like(x) for x in ins] if not packed else None
            pcs = [torch.empty(rows, cols // 2, dtype=torch.uint8, device=dev) for _ in ins] if packed else None
            pes = [torch.empty(rows, cols // c['block_size'], dtype=torch.int8, device=dev) for _ in ins] if packed else None

            def run_rot():
                for i in range(L):
                    r = i % R
                    native.quantize_nm(ins[r], c['block_size'], c['mant_bits'], c['epsilon'], N=N_, M=M_, sparsify_first=(c['first'] == 's'),
                                       want_deq=not packed, code_bits=4 if packed else 0, want_exp=bool(packed),
                                       out=outs[r] if outs else None, codes_out=pcs[r] if pcs else None, exps_out=pes[r] if pes else None)
            run_rot(); torch.cuda.synchronize()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                run_rot()
            g2.replay(); torch.cuda.synchronize()
            ts2 = []
            for _ in range(args.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
                ts2.append(e0.elapsed_time(e1) * 1e3 / L)
            us_rot = statistics.median(ts2)
            del g2, outs, pcs, pes
        results.append(dict(case=name, rows=rows, cols=cols, dtype=dname, us_per_call=us, elems_per_s=numel / us * 1e6,
                            algorithmic_bytes_per_elem=bpe, achieved_GBps=gbps, frac_of_8TBps=gbps / 8000, launch=mode, rotating_inputs=R,
                            us_per_call_rotating_outputs=us_rot, frac_of_8TBps_rotating_outputs=(numel * bpe / us_rot / 8e6) if us_rot else None))
        rot = f"  | rotating outputs {us_rot:7.2f} us ({numel * bpe / us_rot / 80e3:5.1f}%)" if us_rot else ""
        print(f"{name:78s} {us:9.2f} us  {numel/us/1e3:8.1f} Gelem/s  {gbps:7.0f} GB/s ({gbps/80:5.1f}%)  {mode}{rot}", flush=True)
        del ins
        torch.cuda.empty_cache()
    if args.models:
        results += model_rows(args, dev)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
