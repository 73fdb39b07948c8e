#!/usr/bin/env python3
"""bench.py -- throughput of the BFP quantize + sparsify hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (reference float_to_bfp_blocked, bfp_ops.py:124-149) over one
synthetic LLaMA-7B down_proj weight [4096, 11008] bf16: 2:4 magnitude pruning then HBFP4 (sign + 3
mantissa bits, shared exponent per block of 64), drop-in mode (dequantised bf16 tensor out) -- the
configuration BASELINE.json's metric is quoted on.

N = 1: steps rotate over `--rotate` distinct input/output buffer pairs (default 8 x 90 MB in + 8 x 90 MB
out) so reads come from HBM, not from the 256 MB Infinity Cache, and the K timed launches are captured in
one hipGraph (--eager times the plain Python call path instead).

N > 1 (one rank per GPU; `python bench.py --gpus N` starts torch.distributed.run itself when it was not
launched by it): the north-star form -- ONE [4096, 11008] weight row-sharded over the N ranks (rank r owns
rows [r*4096/N, (r+1)*4096/N)), every rank runs the same kernel on its slab and an RCCL all-gather
reassembles the whole fake-quantised tensor on every rank: STRONG scaling, `value` = elements of the whole
tensor / time of (kernel + all-gather).  The packed form of the gather (4-bit codes + int8 exponents,
0.516 B/element on the wire instead of 2) is timed next to it and reported under "packed".  Before timing,
rank 0 checks the gathered tensor bit for bit against an un-sharded run of the same kernel.
--weak keeps the round-1 measurement: every rank its own [4096, 11008] slab, no collective.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--cols", type=int, default=11008)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--mant-bits", type=int, default=3)
    ap.add_argument("--block", type=int, default=64)
    ap.add_argument("--nm", default="2:4", help="N:M, or 0:0 for dense")
    ap.add_argument("--first", default="s", choices=["s", "q"])
    ap.add_argument("--mode", default="dropin", choices=["dropin", "packed", "both"], help="N = 1 / --weak: which outputs the kernel writes")
    ap.add_argument("--rotate", type=int, default=8)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank its own [rows, cols] slab, no collective")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the sub-records (packed, cfg4, two_lanes, model passes): for profiler runs, "
                                                                 "so that the kernel statistics hold the timed region's launches only")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real multi-GPU; gloo only to rehearse N>1 on one GPU")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD process (nothing in this
    process has touched the GPU yet) and pass its output through"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def cpu_threads():
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return min(n, int(os.environ.get("BFPQ_CPU_THREADS", "16")))   # a 1-GPU box's CPU share is 16 cores


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(args, dtype, cfg_kwargs):
    """The reference's own CPU path does not travel to the GPU box; what is timed beside the GPU number is
    (a) oracle/torch_restatement.py -- the same ATen op sequence as bfp_ops.py:29-149, pinned bit for bit to the
        reference's fixtures (tests/test_oracle_golden.py) -- i.e. what the reference's pure-PyTorch path costs here, and
    (b) oracle/bfp_oracle.c, the C + OpenMP port (the faster, conservative comparison).  Baselines only."""
    import torch
    from oracle import oracle as O
    from oracle import torch_restatement as R
    threads = cpu_threads()
    os.environ["OMP_NUM_THREADS"] = str(threads)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1234)
    w = (torch.randn(args.rows, args.cols, generator=g) * 0.02).to(dtype)

    def timed(fn, budget):
        fn(w[:64])                                                 # build + warm
        n, t0 = 0, time.perf_counter()
        while True:
            fn(w)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget or n >= 50:
                return n, dt

    n, dt = timed(lambda x: R.fake_quantize(x, **cfg_kwargs, identifier='w'), args.cpu_seconds * 0.6)
    n2, dt2 = timed(lambda x: O.float_to_bfp_blocked(x, **cfg_kwargs, identifier='w'), args.cpu_seconds * 0.4)
    what = f"[{args.rows},{args.cols}] {args.dtype}"
    return {"value": w.numel() * n / dt, "unit": "elems/s", "cores": threads, "kind": "port",
            "port_of": "pure-torch op-for-op restatement of bfp_ops.py:29-149 (oracle/torch_restatement.py: abs/max/log2/ceil/pow/"
                       "div/round/mul/min/max, topk/full/scatter_/where), bit-identical to the reference's fixtures",
            "cpu_model": cpu_model(), "torch_threads": threads,
            "sample": f"{n} full passes over the same {what} workload in {dt:.1f} s",
            "c_port": {"value": w.numel() * n2 / dt2, "unit": "elems/s", "cores": threads,
                       "sample": f"{n2} full passes, oracle/bfp_oracle.c (C + OpenMP, {threads} threads), {dt2:.1f} s"}}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the engine has no CPU path)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import quantization_sparsity_interplay_amd as pkg
    from quantization_sparsity_interplay_amd import native
    pkg.load_library()                                            # fail loudly if the HIP library is missing

    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    N, M = (int(v) for v in args.nm.split(":"))
    N, M = (N, M) if N < M else (0, 0)
    esize = 4 if dtype == torch.float32 else 2
    strong = world > 1 and not args.weak
    pack_bits = 4 if args.mant_bits <= 3 else 8 if args.mant_bits <= 7 else 16
    R = max(1, args.rotate)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def quantize(x, out=None, packed=False, codes_out=None, exps_out=None):
        return native.quantize_nm(x, args.block, args.mant_bits, 1e-8, N=N, M=M, sparsify_first=(args.first == "s"),
                                  want_deq=not packed, code_bits=pack_bits if packed else 0, want_exp=packed,
                                  out=out, codes_out=codes_out, exps_out=exps_out)

    def timed_loop(fn, steps, warmup, graph, repeats=1):
        """(wall seconds, HIP-event milliseconds) of `steps` calls of fn(i), barrier + synchronize on both sides.
        repeats > 1 (sub-records only, hipGraph): the median of that many timed replays -- a single replay of a few
        milliseconds right after an idle gap is also a measurement of the clocks coming back up."""
        for i in range(warmup):
            fn(i)
        sync_all()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(steps):
                    fn(i)
            g.replay()                                              # untimed: upload + first replay
            sync_all()
            if repeats > 1:
                walls, evs = [], []
                for _ in range(repeats):
                    t0 = time.perf_counter()
                    ev0.record()
                    g.replay()
                    ev1.record()
                    sync_all()
                    walls.append(time.perf_counter() - t0)
                    evs.append(ev0.elapsed_time(ev1))
                return sorted(walls)[repeats // 2], sorted(evs)[repeats // 2]
            t0 = time.perf_counter()
            ev0.record()
            g.replay()
            ev1.record()
        else:
            t0 = time.perf_counter()
            ev0.record()
            for i in range(steps):
                fn(i)
            ev1.record()
        sync_all()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(x):
        if world == 1:
            return [x]
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        out = torch.empty(world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return [float(v) for v in out.cpu()]

    extra = {}
    if not strong:
        # ---- one rank = one whole [rows, cols] tensor (N = 1: the headline; --weak: N independent slabs) -----------
        want_deq = args.mode in ("dropin", "both")
        code_bits = 0 if args.mode == "dropin" else pack_bits
        want_exp = args.mode != "dropin"
        numel = args.rows * args.cols
        ins, outs = [], []
        for r in range(R):
            g = torch.Generator().manual_seed(1234 + r + 1000 * rank)
            ins.append((torch.randn(args.rows, args.cols, generator=g) * 0.02).to(dtype).to(dev))
            outs.append(torch.empty(args.rows, args.cols, dtype=dtype, device=dev) if want_deq else None)

        def step(i):
            r = i % R
            return native.quantize_nm(ins[r], args.block, args.mant_bits, 1e-8, N=N, M=M, sparsify_first=(args.first == "s"),
                                      want_deq=want_deq, code_bits=code_bits, want_exp=want_exp, out=outs[r])

        use_graph = not args.eager and args.mode == "dropin"
        wall, ev_ms = timed_loop(step, args.steps, args.warmup, use_graph)
        wall = max_over_ranks(wall)
        if world == 1 and args.mode == "dropin" and not args.eager and args.block == 64 and pack_bits == 4 and esize == 2 and not args.headline_only:
            # ---- two more measurements of the same call path on the same rotating inputs (sub-records of the line) ----------
            # (1) north_star's "packed int4 stores": the kernel writes 4-bit codes + int8 exponents and no dequantised tensor
            short = max(40, args.steps // 4)
            pcs = [torch.empty(args.rows, args.cols // 2, dtype=torch.uint8, device=dev) for _ in range(R)]
            pes = [torch.empty(args.rows, args.cols // args.block, dtype=torch.int8, device=dev) for _ in range(R)]

            def packed_step(i):
                r = i % R
                return native.quantize_nm(ins[r], args.block, args.mant_bits, 1e-8, N=N, M=M, sparsify_first=(args.first == "s"),
                                          want_deq=False, code_bits=4, want_exp=True, codes_out=pcs[r], exps_out=pes[r])
            _, p_ms = timed_loop(packed_step, short, 10, True, repeats=5)
            p_us = p_ms * 1e3 / short
            p_bytes = numel * esize + numel // 2 + numel // args.block
            extra["packed"] = {"us": p_us, "GB/s": p_bytes / p_us / 1e3, "frac": p_bytes / p_us / 1e3 / HBM_PEAK_GBPS,
                               "algorithmic_bytes_per_launch": p_bytes, "launch": "hipGraph, median of 5 replays", "rotating_buffers": R,
                               "what": "same inputs, same 2:4 -> HBFP4 arithmetic; output = 4-bit two's-complement codes + one int8 shared exponent "
                                       "per block of 64 (2 + 0.5 + 1/64 B per element), k_fused_flat<.., PACK = 4>"}
            del pcs, pes
            # (2) BASELINE config 4's single-tensor form: LLaMA-13B q_proj [5120,5120], 50 % unstructured pruning then HBFP4
            from quantization_sparsity_interplay_amd.bfp import bfp_ops
            c4 = pkg.BFPConfig.hbfp(args.mant_bits + 1, args.block, w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=0.5,
                                    first='s').to_kwargs()
            u_ins = []
            for r in range(R):
                g = torch.Generator().manual_seed(4321 + r)
                u_ins.append((torch.randn(5120, 5120, generator=g) * 0.02).to(dtype).to(dev))
            u_outs = [torch.empty_like(u) for u in u_ins]
            ws = bfp_ops._workspace(dev)

            def unstructured_step(i):
                r = i % R
                return native.prune_quantize(u_ins[r], u_ins[r].numel() // 2, ws, args.block, args.mant_bits, 1e-8, out=u_outs[r])
            _, u_ms = timed_loop(unstructured_step, short, 10, True, repeats=5)
            u_us = u_ms * 1e3 / short
            un = 5120 * 5120
            extra["cfg4_unstructured"] = {"us": u_us, "elems/s": un / u_us * 1e6,
                                          "frac_two_read": un * 3 * esize / u_us / 1e3 / HBM_PEAK_GBPS, "frac_single_read": un * 2 * esize / u_us / 1e3 / HBM_PEAK_GBPS,
                                          "launch": "hipGraph", "rotating_buffers": R,
                                          "what": f"[5120,5120] {args.dtype}: global 50 % magnitude pruning then HBFP4 block 64 (first='s'), selection launch + fused "
                                                  "prune+quantize launch; two-read figure = 6 B/element (the tensor is read by both launches), single-read = 4 B/element"}
            # ... and the other order (first='q': quantize, then prune the quantized tensor), through the reference-shaped call
            c4q = dict(c4, first='q')

            def unstructured_q_step(i):
                return bfp_ops.float_to_bfp_blocked(u_ins[i % R], **c4q, identifier='w')
            _, uq_ms = timed_loop(unstructured_q_step, short, 10, True, repeats=5)
            uq_us = uq_ms * 1e3 / short
            extra["cfg4_unstructured_q_first"] = {"us": uq_us, "elems/s": un / uq_us * 1e6, "launch": "hipGraph", "rotating_buffers": R,
                                                  "what": f"[5120,5120] {args.dtype}: HBFP4 block 64 then global 50 % magnitude pruning of the QUANTIZED tensor (first='q'): quantize launch, "
                                                          "selection launch on its output, prune launch (10 B/element); float_to_bfp_blocked returns a new tensor per call"}
            del u_ins, u_outs
            # (3) the same launches spread over two lanes (the caller's stream and one side stream: bfpq_fake_quantize_list): what a pass
            #     over many tensors gets per tensor, the tail of one launch running beside the ramp of the next
            fq = native.FastQuant(args.block, args.mant_bits, 1e-8, N, M, args.first == "s")
            reps = 64
            pl = native.PreparedList(fq, [ins[i % R] for i in range(reps)], outs=[outs[i % R] for i in range(reps)])
            rounds = max(1, short // reps)
            _, l_ms = timed_loop(lambda i: pl.run(), rounds, 1, True, repeats=5)
            l_us = l_ms * 1e3 / (rounds * reps)
            extra["two_lanes"] = {"us_per_tensor": l_us, "elems/s": numel / l_us * 1e6, "GB/s": 2 * numel * esize / l_us / 1e3,
                                  "frac": 2 * numel * esize / l_us / 1e3 / HBM_PEAK_GBPS, "launch": "hipGraph", "tensors_per_list_call": reps,
                                  "what": "the headline launches as ONE list call per 64 tensors (same rotating buffers): every tensor still gets its own "
                                          "k_fused_flat launch, alternating between the caller's stream and a side stream; two launches share the chip, so "
                                          "this is a throughput figure, not a per-launch duration"}
            del pl
            # (4) BASELINE configs 3 and 4 whole: every Linear weight of LLaMA-7B (2:4 -> HBFP4) and of LLaMA-13B (50 % unstructured -> HBFP4)
            #     through one prepared list call each (weights drawn on the device, randn * 0.02; outputs bound once)
            def model_pass(shapes, kw, bytes_per_elem):
                gd = torch.Generator(device=dev).manual_seed(1234)
                wl = [(torch.randn(r, c, generator=gd, device=dev) * 0.02).to(dtype) for r, c in shapes]
                prep = bfp_ops.PreparedMany(wl, identifier='w', **kw)
                prep.run(); prep.run()
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); prep.run(); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                ms = sorted(ts)[len(ts) // 2]
                n = sum(w.numel() for w in wl)
                del prep, wl
                torch.cuda.empty_cache()
                return {"ms_per_pass": ms, "tensors": len(shapes), "elems": n, "elems/s": n / ms * 1e3,
                        "frac": n * bytes_per_elem / ms / 1e6 / HBM_PEAK_GBPS, "bytes_per_elem": bytes_per_elem}

            def llama(h, inter, layers):
                return [(h, h)] * 4 * layers + [(inter, h)] * 2 * layers + [(h, inter)] * layers
            try:
                c3 = pkg.BFPConfig.hbfp(args.mant_bits + 1, args.block, w_sparsity=True, N=N, M=M, sparsity_mode='structured', first=args.first).to_kwargs()
                extra["model_pass_cfg3"] = dict(model_pass(llama(4096, 11008, 32), c3, 2 * esize),
                                                what=f"LLaMA-7B, all 224 Linear weights {args.dtype}: {args.nm} -> HBFP{args.mant_bits + 1} block {args.block}, one prepared "
                                                     "list call (bfpq_fake_quantize_list, two lanes), HIP events around the eager call, median of 5")
                extra["model_pass_cfg4"] = dict(model_pass(llama(5120, 13824, 40), c4, 3 * esize),
                                                what=f"LLaMA-13B, all 280 Linear weights {args.dtype}: 50 % unstructured -> HBFP{args.mant_bits + 1} block {args.block}, one "
                                                     "prepared list call (bfpq_prune_quantize_list, four lanes, its own hipGraph); frac on the two-read "
                                                     "figure (6 B per element: the selection and the prune + quantize launch both read the tensor)")
            except Exception as e:                                  # (never at the cost of the main line)
                extra["model_pass_error"] = f"{type(e).__name__}: {e}"
        value = world * numel * args.steps / wall
        kern_us = ev_ms * 1e3 / args.steps
        bytes_per_launch = numel * esize + (numel * esize if want_deq else 0) + (numel * code_bits // 8 if code_bits else 0) + \
            (numel // args.block if (want_exp and args.block) else 0)
        parallelism = f"row-sharded x{world}, no collective (weak scaling)" if world > 1 else "1 GPU"
        launch = "hipGraph" if use_graph else "eager"
        output = args.mode
        rows_per_gpu = args.rows
    else:
        # ---- north-star form: ONE [rows, cols] weight, row-sharded, RCCL all-gather of the result -------------------
        from quantization_sparsity_interplay_amd import dist as qd
        chk = torch.tensor([float(rank + 1)], device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(chk)                                        # did the collective library see `world` ranks?
        extra["rccl_ranks"] = world if abs(float(chk.item()) - world * (world + 1) / 2) < 1e-6 else -1
        extra["collective_backend"] = "RCCL (torch.distributed 'nccl')" if args.backend == "nccl" else args.backend
        if args.rows % world:
            raise SystemExit(f"--rows {args.rows} must divide by the number of ranks ({world}) in the strong-scaling mode")
        lo, hi = qd.row_range(args.rows, world, rank)
        per = hi - lo
        numel = args.rows * args.cols
        fulls = []
        for r in range(R):
            g = torch.Generator().manual_seed(1234 + r)
            fulls.append((torch.randn(args.rows, args.cols, generator=g) * 0.02).to(dtype))     # same bits as the 1-GPU run
        ins = [f[lo:hi].to(dev) for f in fulls]
        slabs = [torch.empty(per, args.cols, dtype=dtype, device=dev) for _ in range(R)]
        wholes = [torch.empty(args.rows, args.cols, dtype=dtype, device=dev) for _ in range(R)]
        nblk = (args.cols + args.block - 1) // args.block
        ccols = (args.cols + 1) // 2 if pack_bits == 4 else args.cols
        cdt = {4: torch.uint8, 8: torch.int8, 16: torch.int16}[pack_bits]
        pc = [torch.empty(per, ccols, dtype=cdt, device=dev) for _ in range(2)]
        pe = [torch.empty(per, nblk, dtype=torch.int8, device=dev) for _ in range(2)]
        wc = [torch.empty(args.rows, ccols, dtype=cdt, device=dev) for _ in range(2)]
        we = [torch.empty(args.rows, nblk, dtype=torch.int8, device=dev) for _ in range(2)]

        def kernel_only(i):
            quantize(ins[i % R], out=slabs[i % R])

        def gather_only(i):
            qd.all_gather_into(wholes[i % R], slabs[i % R])

        def step(i):
            kernel_only(i)
            gather_only(i)

        def packed_step(i):
            quantize(ins[i % R], packed=True, codes_out=pc[i % 2], exps_out=pe[i % 2])
            qd.all_gather_into(wc[i % 2], pc[i % 2])
            qd.all_gather_into(we[i % 2], pe[i % 2])

        def packed_gather_only(i):
            qd.all_gather_into(wc[i % 2], pc[i % 2])
            qd.all_gather_into(we[i % 2], pe[i % 2])

        # parity of the sharded form: gathered == un-sharded, bit for bit (rank 0 has the whole input)
        step(0)
        packed_step(0)
        torch.cuda.synchronize()
        if rank == 0:
            ref = quantize(fulls[0].to(dev))[0]
            _, rc, re = quantize(fulls[0].to(dev), packed=True)
            extra["gathered_equals_single_gpu"] = bool(torch.equal(wholes[0].view(torch.int16 if esize == 2 else torch.int32),
                                                                   ref.view(torch.int16 if esize == 2 else torch.int32)))
            extra["gathered_packed_equals_single_gpu"] = bool(torch.equal(wc[0], rc) and torch.equal(we[0], re))
            del ref, rc, re

        def overlapped_step(i):
            # kernel per row chunk on the main stream, the chunk's all-gather on the side stream while the next chunk is computed
            qd.gather_overlapped(ins[i % R], args.rows, lambda p: quantize(p)[0], chunks=4, out=wholes[i % R])

        overlapped_step(0)
        torch.cuda.synchronize()
        if rank == 0:
            extra["overlapped_equals_single_gpu"] = bool(torch.equal(wholes[0].view(torch.int16 if esize == 2 else torch.int32),
                                                                     quantize(fulls[0].to(dev))[0].view(torch.int16 if esize == 2 else torch.int32)))
        short = max(20, args.steps // 4)
        owall, _ = timed_loop(overlapped_step, short, 5, False)
        owall = max_over_ranks(owall)
        extra["overlapped"] = {"value": numel * short / owall, "unit": "elems/s", "ms_per_step": owall * 1e3 / short, "chunks": 4,
                               "what": "dist.gather_overlapped: the slab in 4 row chunks, chunk i's all_gather_into_tensor (persistent side stream, "
                                       "contiguous staging buffer + one strided copy) beside the kernel of chunk i + 1"}
        # every rank a whole [rows, cols] tensor of its own, no collective (the --weak form, here as a sub-record: rows are
        # independent units, so this is what the path does when the caller does NOT need the tensor whole on every rank)
        w_ins = [f.to(dev) for f in fulls[:min(R, 4)]]
        w_outs = [torch.empty_like(x) for x in w_ins]

        def weak_step(i):
            quantize(w_ins[i % len(w_ins)], out=w_outs[i % len(w_ins)])
        wwall, _ = timed_loop(weak_step, short, 5, False)
        wwall = max_over_ranks(wwall)
        extra["weak_no_collective"] = {"value": world * numel * short / wwall, "unit": "elems/s", "ms_per_step": wwall * 1e3 / short,
                                       "what": f"every rank quantizes its own whole [{args.rows},{args.cols}] tensor, no collective "
                                               "(per-GPU work fixed: weak scaling of the path itself), eager"}
        del w_ins, w_outs
        # BASELINE config 4 as specified, at a reduced list: 16 LLaMA-13B q_proj weights [5120,5120], 50 % unstructured -> HBFP4, every rank
        # its row slab of every weight.  lanes = 1: tensor after tensor, one histogram all-gather each; lanes = 4: dealt to four streams,
        # one tensor's exchange beside the others' kernels; one_exchange: all 16 histograms first, ONE all-gather of their folded copies
        # (dist.float_to_bfp_blocked_many_sharded)
        if 5120 % world == 0 and esize == 2:
            gq = torch.Generator(device=dev).manual_seed(4321)
            q_slabs = [(torch.randn(5120 // world, 5120, generator=gq, device=dev) * 0.02).to(dtype) for _ in range(16)]
            c4 = pkg.BFPConfig.hbfp(args.mant_bits + 1, args.block, w_sparsity=True, sparsity_mode='unstructured', sparsity_frac=0.5, first='s').to_kwargs()
            rec = {"tensors": 16, "shape": [5120, 5120], "what": "16 x [5120,5120] row-sharded, 50 % unstructured -> HBFP4 (first = 's'), no result gather: "
                                                                   "selection launch, histogram all-gather, resolve launch, prune + quantize launch per tensor"}
            for ln in (1, 4):
                def list_step(i, ln=ln):
                    qd.float_to_bfp_blocked_many_sharded(q_slabs, [5120] * 16, identifier='w', lanes=ln, exchange='tensor', **c4)
                lwall, _ = timed_loop(list_step, 5, 2, False)
                rec[f"ms_per_pass_lanes{ln}"] = max_over_ranks(lwall) * 1e3 / 5

            def one_exchange_step(i):
                qd.float_to_bfp_blocked_many_sharded(q_slabs, [5120] * 16, identifier='w', exchange='list', group_size=16, **c4)
            lwall, _ = timed_loop(one_exchange_step, 5, 2, False)
            rec["ms_per_pass_one_exchange"] = max_over_ranks(lwall) * 1e3 / 5            # ONE histogram all-gather for all 16 tensors
            rec["elems/s_one_exchange"] = 16 * 5120 * 5120 / rec["ms_per_pass_one_exchange"] * 1e3
            extra["cfg4_sharded_list"] = rec
            del q_slabs
        _, k_ms = timed_loop(kernel_only, short, 5, False)
        _, c_ms = timed_loop(gather_only, short, 5, False)
        _, pcoll_ms = timed_loop(packed_gather_only, short, 5, False)
        pwall, _ = timed_loop(packed_step, short, 5, False)
        pwall = max_over_ranks(pwall)
        wall, ev_ms = timed_loop(step, args.steps, args.warmup, False)
        wall = max_over_ranks(wall)
        value = numel * args.steps / wall
        kern_us = k_ms * 1e3 / short
        extra["per_rank_kernel_us"] = all_ranks(kern_us)
        extra["per_rank_collective_us"] = all_ranks(c_ms * 1e3 / short)
        extra["packed"] = {"value": numel * short / pwall, "unit": "elems/s", "ms_per_step": pwall * 1e3 / short,
                           "wire_bytes_per_rank": per * (ccols * (1 if pack_bits <= 8 else 2) + nblk),
                           "per_rank_collective_us": all_ranks(pcoll_ms * 1e3 / short),
                           "what": f"kernel writes {pack_bits}-bit codes + int8 exponents, two all-gathers reassemble them"}
        extra["wire_bytes_per_rank"] = per * args.cols * esize
        bytes_per_launch = 2 * per * args.cols * esize                # the slab kernel: read once, write once
        parallelism = f"one [{args.rows},{args.cols}] weight row-sharded x{world} + RCCL all-gather of the {args.dtype} result (strong scaling)"
        launch = "eager (kernel + collective per step)"
        output = "dropin, gathered whole on every rank"
        rows_per_gpu = per

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and not strong:
            try:
                tj = json.load(open(tpath))
                key = f"{args.rows}x{args.cols}_{args.dtype}_m{args.mant_bits}_b{args.block}_{args.nm}_{args.first}_{args.mode}"
                if key in tj:
                    traffic, traffic_src = tj[key]["hbm_bytes_per_launch"], tj[key].get("source")
            except Exception:
                pass
        cfg_kwargs = pkg.BFPConfig.hbfp(args.mant_bits + 1, args.block, w_sparsity=M > 0, N=N, M=M,
                                        sparsity_mode='structured', first=args.first).to_kwargs()
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, dtype, cfg_kwargs)
        line = {
            "metric": "weight elems/sec quantized+sparsified (BFP-int4, 2:4) on 4096x11008; % HBM roofline",
            "value": value, "unit": "elems/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": ("strong" if strong else "weak") if world > 1 else None, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"LLaMA-7B down_proj weight [{args.rows},{args.cols}] {args.dtype} -> "
                                   f"{args.nm} magnitude pruning ({'sparsify->quantize' if args.first == 's' else 'quantize->sparsify'}) "
                                   f"+ HBFP{args.mant_bits + 1} (mant_bits={args.mant_bits}, block={args.block}), round-half-even",
                       "output": output, "rows_per_gpu": rows_per_gpu, "cols": args.cols,
                       "parallelism": parallelism, "launch": launch, "rotating_buffers": R},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "kernel": "k_fused_flat",
                         "avg_launch_us": kern_us, "traffic_source": traffic_src},
            "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
