#!/usr/bin/env python3
"""bench.py -- throughput of the BFP quantize + sparsify hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (reference float_to_bfp_blocked, bfp_ops.py:124-149) over one
synthetic LLaMA-7B down_proj weight [4096, 11008] bf16: 2:4 magnitude pruning then HBFP4 (sign + 3
mantissa bits, shared exponent per block of 64), drop-in mode (dequantised bf16 tensor out) -- the
configuration BASELINE.json's metric is quoted on.  Steps rotate over `--rotate` distinct input/output
buffer pairs (default 8 x 90 MB in + 8 x 90 MB out) so reads come from HBM, not from the 256 MB
Infinity Cache.  The K timed launches are captured in one hipGraph (the per-launch host cost of the
Python boundary is otherwise comparable to the ~40 us kernel); --eager times the plain call path.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): rows are the sharding unit, every
rank quantizes its own [4096, 11008] slab (weak scaling, no data-path collective);
--allgather additionally reassembles the packed result on every rank with an RCCL all-gather.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

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
    ap.add_argument("--mode", default="dropin", choices=["dropin", "packed", "both"])
    ap.add_argument("--rotate", type=int, default=8)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--allgather", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real multi-GPU; gloo only to rehearse N>1 on one GPU")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def cpu_baseline(args, dtype, N, M, cfg_kwargs):
    """the CPU oracle (a C/OpenMP port of the reference algorithm, oracle/bfp_oracle.c) timed on this
    box's host cores on the same workload; baseline only."""
    from oracle import oracle as O
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = min(threads, int(os.environ.get("BFPQ_CPU_THREADS", "16")))   # a 1-GPU box's CPU share is 16 cores
    os.environ["OMP_NUM_THREADS"] = str(threads)
    g = torch.Generator().manual_seed(1234)
    w = (torch.randn(args.rows, args.cols, generator=g) * 0.02).to(dtype)
    O.float_to_bfp_blocked(w[:64], **cfg_kwargs, identifier='w')          # build + warm
    t_end = time.perf_counter() + args.cpu_seconds
    n = 0
    t0 = time.perf_counter()
    while True:
        O.float_to_bfp_blocked(w, **cfg_kwargs, identifier='w')
        n += 1
        if time.perf_counter() >= t_end or n >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": w.numel() * n / dt, "unit": "elems/s", "cores": threads, "kind": "port",
            "sample": f"{n} full passes over the same [{args.rows},{args.cols}] {args.dtype} workload, "
                      f"oracle/bfp_oracle.c (C + OpenMP, {threads} threads), {dt:.1f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for N>1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
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
    esize = 4 if dtype == torch.float32 else 2
    numel = args.rows * args.cols
    want_deq = args.mode in ("dropin", "both")
    code_bits = 0 if args.mode == "dropin" else (4 if args.mant_bits <= 3 else 8 if args.mant_bits <= 7 else 16)
    want_exp = args.mode != "dropin"

    # synthetic inputs: seed 1234 (+ buffer index, + rank), randn * 0.02, generated on CPU then copied
    R = max(1, args.rotate)
    ins, outs = [], []
    for r in range(R):
        g = torch.Generator().manual_seed(1234 + r + 1000 * rank)
        ins.append((torch.randn(args.rows, args.cols, generator=g) * 0.02).to(dtype).to(dev))
        outs.append(torch.empty(args.rows, args.cols, dtype=dtype, device=dev) if want_deq else None)

    gather_buf = None
    if args.allgather and world > 1:
        gather_buf = torch.empty(world * args.rows, args.cols, dtype=dtype, device=dev)

    def step(i):
        r = i % R
        res = native.quantize_nm(ins[r], args.block, args.mant_bits, 1e-8, N=N if N < M else 0, M=M if N < M else 0,
                                 sparsify_first=(args.first == "s"), want_deq=want_deq, code_bits=code_bits,
                                 want_exp=want_exp, out=outs[r])
        if gather_buf is not None:
            dist.all_gather_into_tensor(gather_buf, res[0])
        return res

    if not pkg.load_library().bfpq_is_fused(args.rows, args.cols, native.DTYPE_CODE[dtype], args.block, N, M):
        print("note: this shape takes the general (multi-launch) path", file=sys.stderr)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()

    use_graph = not args.eager and gather_buf is None and args.mode == "dropin"
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if use_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(args.steps):
                step(i)
        graph.replay()                                              # untimed: upload + first replay
        sync_all()
        t0 = time.perf_counter()
        ev0.record()
        graph.replay()
        ev1.record()
        sync_all()
        t1 = time.perf_counter()
    else:
        sync_all()
        t0 = time.perf_counter()
        ev0.record()
        for i in range(args.steps):
            step(i)
        ev1.record()
        sync_all()
        t1 = time.perf_counter()

    wall = t1 - t0
    ev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = world * numel * args.steps / wall
        # algorithmic bytes per launch: read the tensor once, write each requested output once
        nblk = numel // args.block if args.block else 0
        bytes_per_launch = numel * esize
        if want_deq:
            bytes_per_launch += numel * esize
        if code_bits:
            bytes_per_launch += numel * code_bits // 8
        if want_exp:
            bytes_per_launch += nblk
        kern_us = ev_ms * 1e3 / args.steps
        achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.rows}x{args.cols}_{args.dtype}_m{args.mant_bits}_b{args.block}_{args.nm}_{args.first}_{args.mode}"
                if key in tj:
                    traffic, traffic_src = tj[key]["hbm_bytes_per_launch"], tj[key].get("source")
            except Exception:
                pass
        cfg_kwargs = pkg.BFPConfig.hbfp(args.mant_bits + 1, args.block, w_sparsity=(N < M and M > 0), N=N, M=M,
                                        sparsity_mode='structured', first=args.first).to_kwargs()
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, dtype, N, M, cfg_kwargs)
        line = {
            "metric": "weight elems/sec quantized+sparsified (BFP-int4, 2:4) on 4096x11008; % HBM roofline",
            "value": value, "unit": "elems/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"LLaMA-7B down_proj weight [{args.rows},{args.cols}] {args.dtype} -> "
                                   f"{args.nm} magnitude pruning ({'sparsify->quantize' if args.first == 's' else 'quantize->sparsify'}) "
                                   f"+ HBFP{args.mant_bits + 1} (mant_bits={args.mant_bits}, block={args.block}), round-half-even",
                       "output": args.mode, "rows_per_gpu": args.rows, "cols": args.cols,
                       "parallelism": f"row-sharded x{world}" + (" + RCCL all-gather" if gather_buf is not None else ""),
                       "launch": "hipGraph" if use_graph else "eager", "rotating_buffers": R},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "kernel": "k_fused_flat",
                         "avg_launch_us": kern_us, "traffic_source": traffic_src},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
