#!/usr/bin/env python3
"""Headline benchmark: molecules/sec, fwd+bwd of ONE TripletMessage(60, 4, heads=3) layer on an
ESOL-shaped batch of 1024 molecules (BASELINE.json configs[1]), fp32, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = forward + backward of the layer over one batch (loss = <out, cotangent>), gradients of the
5 parameters gathered into one flat bucket; with N>1 every rank owns its own 1024-molecule batch
(weak scaling, graphs shard with no data-path collective) and the step ends with ONE RCCL
all-reduce of that bucket.  The step is captured once into a hipGraph (launch-bound regime:
~20 kernels of 2-6 us) and replayed; ``--no-graph`` times eager launches instead.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant hand-written kernel: algorithmic bytes / avg launch duration (HIP events on
                the launching stream, back-to-back launches) vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (reference-shaped port, oracle/glam_oracle.py) timed on this host
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md (spec); 6290 GB/s measured-achievable


def algorithmic_bytes(N, E, H=3, C=60, De=4):
    """Compulsory HBM bytes per launch, fp32 + int32 CSR, every tensor once."""
    HC = H * C
    fwd = 4 * (2 * N * HC + E * De + 2 * E + (N + 1) + 2 * N * H + 2 * N * H)
    # B1 (by target): xw gather, aggr, d_aggr rows, edge_attr, src/eid, rowptr, a_ij, stats in;
    #                 alpha_e/dpre_e [E,H] and d_a_i out
    b1 = 4 * (3 * N * HC + E * De + 2 * E + (N + 1) + 2 * N * H + 2 * N * H + 2 * E * H + N * H)
    # B2 (by source): d_aggr gather, edge_attr, dst/eid, colptr, alpha_e/dpre_e in; d_xw, d_a_j out
    b2 = 4 * (2 * N * HC + E * De + 2 * E + (N + 1) + 2 * E * H + N * H)
    return {"k_triplet_fwd": fwd, "k_triplet_bwd_dst": b1, "k_triplet_bwd_src": b2}


def pmc_traffic(kernel, N):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r1_hbm_traffic_pmc.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this script, reads doubled per the gfx950 note of
    MI355X_MICROARCH.md §HBM).  Matched on the launch's thread count; None when no matching measurement is committed."""
    path = os.path.join(ROOT, "profiles", "r1_hbm_traffic_pmc.json")
    try:
        rows = json.load(open(path))
    except OSError:
        return None
    want = f"grid={(N + 15) // 16 * 256}"
    for key, row in rows.items():
        if kernel in key and key.endswith(want):
            return row["hbm_bytes"]
    return None


def time_kernels(conv, batch, x, reps=200):
    """Average duration of each hand-written aggregate kernel: `reps` back-to-back launches between two
    HIP events on the stream the kernels are launched on (torch's current stream)."""
    from glam_amd import _lib, ops
    lib = _lib.load()
    p, st = _lib.ptr, _lib.stream
    N, E = x.size(0), batch.edge_index.size(1)
    gi = ops.graph_index(batch.edge_index, N)
    colptr, dst, eid_t = gi.transpose()
    with torch.no_grad():
        Wn, Wa, We, M, Ws, Cp, Dp = conv._staged_weights()
        xw, a_ij = (x @ Wn).contiguous(), (x @ Wa).contiguous()
        ea = batch.edge_attr
        H = conv.heads
        aggr, stats = torch.empty(N, H * Cp, device=x.device), torch.empty(N, 8, device=x.device)
        d_aggr = torch.randn(N, H * Cp, device=x.device)
        d_xw, d_a = torch.empty_like(xw), torch.empty_like(a_ij)
        d_we, d_M = torch.empty_like(We), torch.empty_like(M)
        ws = torch.empty(lib.glam_triplet_bwd_workspace_bytes(N, E, H, Cp, Dp), dtype=torch.uint8, device=x.device)

    def fwd():
        lib.glam_triplet_fwd(p(xw), p(a_ij), p(ea), p(We), p(M), p(gi.rowptr), p(gi.src), p(gi.eid), N, E, H, Cp, Dp, 1,
                             0.2, p(aggr), p(stats), st())

    def bwd():
        lib.glam_triplet_bwd(p(xw), p(a_ij), p(ea), p(We), p(M), p(aggr), p(stats), p(d_aggr), p(gi.rowptr), p(gi.src),
                             p(gi.eid), p(colptr), p(dst), p(eid_t), N, E, H, Cp, Dp, 1, 0.2, p(d_xw), p(d_a), p(d_we),
                             p(d_M), None, p(ws), ws.numel(), st())

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps   # us per call

    t_fwd = timed(fwd)
    t_bwd = timed(bwd)       # B1 + partial reduce + B2 (split by the rocprof trace in profiles/)
    return {"k_triplet_fwd": t_fwd, "triplet_bwd(B1+reduce+B2)": t_bwd}


def cpu_baseline(batch_cpu, conv_cpu, x_cpu, cot_cpu, budget_s):
    import oracle.glam_oracle as O   # checker only: timed here as the CPU baseline, never shipped
    ps = [p.detach().clone().requires_grad_(True) for p in conv_cpu.parameters()]
    x = x_cpu.clone().requires_grad_(True)

    def step():
        out = O.triplet_message(x, batch_cpu.edge_index, batch_cpu.edge_attr, *ps)
        torch.autograd.grad((out * cot_cpu).sum(), [x] + ps)

    # torch's CPU kernels on this path (index_select / scatter / small GEMMs) do not scale to every core of a big host:
    # time a few thread counts inside the budget and report the FASTEST as the baseline (with the count it used).
    avail = torch.get_num_threads()
    counts = sorted({c for c in (8, 32, avail) if c <= avail})
    B = int(batch_cpu.batch[-1]) + 1
    runs = {}
    for c in counts:
        torch.set_num_threads(c)
        for _ in range(2):
            step()
        reps, t0 = 0, time.perf_counter()
        while True:
            step()
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s / len(counts) or reps >= 200:
                break
        runs[c] = (reps, dt)
    torch.set_num_threads(avail)
    best = min(runs, key=lambda c: runs[c][1] / runs[c][0])
    reps, dt = runs[best]
    return {"value": B * reps / dt, "unit": "molecules/s", "cores": best, "kind": "port",
            "sample": f"{reps} fwd+bwd steps of the same B={B} batch in {dt:.1f} s (oracle/glam_oracle.py, torch CPU fp32); "
                      f"fastest of {counts} threads on a {os.cpu_count()}-cpu host",
            "ms_per_step": dt / reps * 1e3,
            "ms_per_step_by_threads": {str(c): runs[c][1] / runs[c][0] * 1e3 for c in counts}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1024, help="molecules per GPU per step")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget (0 = skip)")
    ap.add_argument("--large-batch", type=int, default=16384, help="extra roofline point beyond the LLC (0 = skip)")
    ap.add_argument("--storage", choices=["fp32", "bf16"], default="fp32",
                    help="storage of the gathered rows xw (bf16 = BASELINE configs[2] mode: bf16 rows, fp32 arithmetic; "
                         "the headline metric is fp32)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # Functional check of the multi-rank path on a single-GPU box (not a measurement): GLAM_BENCH_SHARE_GPU=1 puts every rank
    # on device 0 and uses gloo, since RCCL refuses two ranks on one device.
    share = os.environ.get("GLAM_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI

    from glam_amd import layer, ops
    from glam_amd.data import synth_batch
    ops.FEATURE_STORAGE = args.storage
    from glam_amd.parallel import broadcast_parameters, flat_view

    B, C, De, H = args.batch, 60, 4, 3
    batch_cpu = synth_batch(B, seed=rank)                      # every rank: its own B molecules (weak scaling)
    torch.manual_seed(0)
    conv_cpu = layer.TripletMessage(C, De, heads=H)
    N, E = batch_cpu.x.size(0), batch_cpu.edge_index.size(1)
    g = torch.Generator().manual_seed(100 + rank)
    x_cpu = torch.randn(N, C, generator=g)                     # hidden state h ~ N(0,1)
    cot_cpu = torch.randn(N, C, generator=g)

    import copy
    conv = copy.deepcopy(conv_cpu).to(dev)
    broadcast_parameters(conv)
    batch = batch_cpu.to(dev)
    x = x_cpu.to(dev).requires_grad_(True)
    cot = cot_cpu.to(dev)
    params = list(conv.parameters())
    n_param = sum(p.numel() for p in params)
    flat = torch.zeros(n_param, device=dev)                    # gradient bucket when the grads are not one already
    live = {}

    def body():
        out = conv(x, batch.edge_index, batch.edge_attr)
        grads = torch.autograd.grad(out, params + [x], grad_outputs=cot)   # loss = <out, cot>
        bucket = flat_view(grads[:-1])          # the fused layer hands its 5 parameter gradients back as one buffer
        if bucket is None:
            bucket = torch.cat([g_.reshape(-1) for g_ in grads[:-1]], out=flat)
        live["bucket"], live["d_x"] = bucket, grads[-1]

    # warm-up on a side stream (stages the CSR + its transpose, which sync once per new batch)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()

    def step():
        if graph is not None:
            graph.replay()
        else:
            body()
        if world > 1:
            dist.all_reduce(live["bucket"])                    # ONE bucket, ONE RCCL call per step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    result = {
        "metric": "molecules/sec fwd+bwd on ESOL-shaped batches", "value": value, "unit": "molecules/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.storage == "fp32" else "bf16 rows / f32 arithmetic", "data": "synthetic",
        "config": {"workload": f"ESOL-shaped batch={B}/GPU (N={N} atoms, E={E} directed bonds), single "
                               f"TripletMessage({C},{De},heads={H}) layer fwd+bwd, fp32",
                   "launch": "eager" if graph is None else "hipGraph replay", "parallelism": f"dp{world}",
                   "global_batch": B * world},
    }

    if rank == 0 and args.storage != "fp32":
        result["roofline"] = None                  # the roofline leg times the fp32 aggregate kernel: headline mode only
        print(json.dumps(result))
    elif rank == 0:
        # ---- roofline of the hand-written kernels (after the timed region, same process/stream) ----
        kt = time_kernels(conv, batch, x.detach())
        ab = algorithmic_bytes(N, E, H, C, De)
        fwd_us = kt["k_triplet_fwd"]
        achieved = ab["k_triplet_fwd"] / (fwd_us * 1e-6) / 1e9
        result["roofline"] = {"kernel": "k_triplet_fwd<3,16,1,4,true> (gather + segment softmax + scatter-add)",
                              "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("k_triplet_fwd", N),
                              "algorithmic_bytes": ab["k_triplet_fwd"], "avg_launch_us": fwd_us,
                              "workload": f"B={B}"}
        bwd_bytes = ab["k_triplet_bwd_dst"] + ab["k_triplet_bwd_src"]
        result["roofline_bwd"] = {"kernels": "k_triplet_bwd_dst + k_reduce_partials + k_triplet_bwd_src",
                                  "algorithmic_bytes": bwd_bytes, "avg_launch_us": kt["triplet_bwd(B1+reduce+B2)"],
                                  "achieved": bwd_bytes / (kt["triplet_bwd(B1+reduce+B2)"] * 1e-6) / 1e9, "unit": "GB/s"}
        if args.large_batch and world == 1:
            big_cpu = synth_batch(args.large_batch, seed=7)
            big = big_cpu.to(dev)
            xb = torch.randn(big.x.size(0), C, device=dev)
            ktb = time_kernels(conv, big, xb, reps=50)
            abb = algorithmic_bytes(big.x.size(0), big.edge_index.size(1), H, C, De)
            a2 = abb["k_triplet_fwd"] / (ktb["k_triplet_fwd"] * 1e-6) / 1e9
            result["roofline_large"] = {"workload": f"B={args.large_batch} (N={big.x.size(0)}, beyond the 256 MiB LLC)",
                                        "kernel": "k_triplet_fwd", "achieved": a2, "frac": a2 / HBM_PEAK_GBS,
                                        "avg_launch_us": ktb["k_triplet_fwd"], "algorithmic_bytes": abb["k_triplet_fwd"],
                                        "bwd_avg_launch_us": ktb["triplet_bwd(B1+reduce+B2)"],
                                        "bwd_achieved": (abb["k_triplet_bwd_dst"] + abb["k_triplet_bwd_src"]) /
                                                        (ktb["triplet_bwd(B1+reduce+B2)"] * 1e-6) / 1e9}
            del big, xb
        if args.cpu_seconds > 0 and world == 1:     # rank 0 at N = 1 only: the other ranks of a multi-GPU run would sit in the barrier
            result["cpu_baseline"] = cpu_baseline(batch_cpu, conv_cpu, x_cpu, cot_cpu, args.cpu_seconds)
            result["gpu_over_cpu"] = value / world / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
