#!/usr/bin/env python3
"""Headline benchmark: molecules/sec, fwd+bwd of ONE TripletMessage(60, 4, heads=3) layer on an
ESOL-shaped batch of 1024 molecules (BASELINE.json configs[1]), fp32, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1: run as written, bench.py starts its own N ranks (fresh child interpreters, one per GPU, started
BEFORE this process touches a GPU); under ``python -m torch.distributed.run --nproc-per-node N ...`` it
joins the ranks the launcher made (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

A step = forward + backward of the layer over one batch (loss = <out, cotangent>), the 5 parameter
gradients in one flat bucket; with N > 1 every rank owns its own 1024-molecule batch (weak scaling,
graphs shard with no data-path collective) and the step ends with ONE RCCL all-reduce of that bucket.
The step — including the all-reduce — is captured once into a hipGraph (launch-bound regime: 8 kernels
of 5-17 us) and replayed; ``--no-graph`` times eager launches instead.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline          the scatter-aggregate kernel THE TIMED STEP LAUNCHES (gather + segment softmax + scatter-add with
                    the update GEMM fused in): algorithmic bytes / its average duration vs the HBM peak
  roofline_kernels  every kernel of the step: durations from per-dispatch begin/end timestamps (glam_prof_*:
                    hipExtLaunchKernel events, the figures a rocprofv3 kernel trace reports) taken in THIS run from
                    eager executions of the very function the graph captured; bytes / flops per DESIGN.md §4
  roofline_inference the forward under torch.no_grad() (the reference's evaluation passes): the inference instantiation of the same kernel
  roofline_isolated the aggregate kernels alone (no fused GEMM; general and software-pipelined forward), SURVEY.md §8(d)'s formula
  roofline_large    the same kernels at B = 16 384 (working set beyond the 256 MiB LLC); frac = the forward scatter-aggregate
  cpu_baseline      the CPU oracle (reference-shaped port, oracle/glam_oracle.py) timed on this host
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

_T_START = time.perf_counter()
import torch
_T_TORCH = time.perf_counter()

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E spec, ~6.29 TB/s measured-achievable; 157.3 TF dense fp32 MFMA
HBM_PEAK_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 6290.0
MFMA_F32_PEAK_TFLOPS = 157.3


def step_kernel_model(N, E, H=3, C=60, De=4):
    """Algorithmic (compulsory) HBM bytes and dense flops per launch of every kernel of the step: fp32 + int32 CSR, every
    tensor once.  Keys are the labels glam_prof_* reports."""
    HC, f = H * C, 4
    img = lambda K, M: ((K + 15) // 16 * 16) * (64 if M <= 64 else 192) * f      # weight image of a k_ts_gemm launch
    agg_fwd = f * (2 * N * HC + E * De + 2 * E + (N + 1) + 2 * N * H + 2 * N * H)        # xw, aggr, edge_attr, src+eid, rowptr, a_ij, stats
    # backward by target: xw (gather) + d_aggr; aggr is not read any more (round 5: sum_e alpha_e d_alpha_e comes from the node's own edges
    # whenever they fit one chunk of four — always, for molecules; GLAM_X3=0 is bit-identical, it does not bring the read back)
    b1 = f * (2 * N * HC + E * De + 2 * E + (N + 1) + 2 * N * H + 2 * N * H + 2 * E * H + N * H)
    b2 = f * (2 * N * HC + E * De + 2 * E + (N + 1) + 2 * E * H + N * H)
    return {
        "k_stage_params": {"bound": "latency", "bytes": f * (C * HC + De * HC + 3 * HC + HC * C + C) + img(C, HC + 8) + img(HC, C) + img(C, HC) + img(HC + 8, C)},
        "k_ts_gemm<12, 4, 4>": {"bound": "mfma", "flops": 2 * N * C * (HC + 8), "bytes": f * N * (C + HC + 8) + img(C, HC + 8),
                                "note": "x @ [W_node | Wa]; a second launch per step (d_out @ W_scale^T, 2*N*C*HC flops) only where B1 has no fused variant"},
        "k_triplet_fwd+update": {"bound": "hbm", "bytes": agg_fwd + f * N * C + img(HC, C), "flops": 2 * N * HC * C},
        "k_triplet_fwd_ws+update": {"bound": "hbm", "bytes": agg_fwd + f * N * C + img(HC, C), "flops": 2 * N * HC * C,
                                    "note": "warp-specialised: 8 producer waves run the software-pipelined aggregate, 4 consumer waves the update GEMM "
                                            "out of an LDS tile ring (csrc/triplet_ws.hip); the op's choice for molecular graphs at every size"},
        # torch.no_grad() (src_1gp/trainer.py:306-327): neither aggr nor stats is stored
        "k_triplet_fwd_ws<inference>+update": {"bound": "hbm", "bytes": agg_fwd - f * (N * HC + 2 * N * H) + f * N * C + img(HC, C), "flops": 2 * N * HC * C,
                                               "note": "the inference instantiation: reads xw (gather), a_ij, edge records; writes out only"},
        "k_triplet_fwd": {"bound": "hbm", "bytes": agg_fwd},
        "k_triplet_fwd_pipe": {"bound": "hbm", "bytes": agg_fwd,
                               "note": "software-pipelined forward aggregate (csrc/triplet_pipe.hip), same arithmetic and SURVEY §8(d) byte model"},
        "k_triplet_bwd_dst": {"bound": "hbm", "bytes": b1},
        # B1 with d_aggr = d_out @ W_scale^T as a per-tile MFMA prologue: d_aggr is written instead of read (same bytes), d_out and the
        # weight image are read in addition
        "d_aggr+k_triplet_bwd_dst": {"bound": "hbm", "bytes": b1 + f * N * C + img(C, HC), "flops": 2 * N * C * HC,
                                     "note": "backward by target with the d_aggr GEMM fused in (one launch and one kernel boundary less)"},
        "k_triplet_bwd_src+dx": {"bound": "hbm", "bytes": b2 + f * N * C + img(HC + 8, C), "flops": 2 * N * (HC + 8) * C},
        "d_aggr+k_triplet_bwd_dst_ws": {"bound": "hbm", "bytes": b1 + f * N * C + img(C, HC), "flops": 2 * N * C * HC,
                                        "note": "warp-specialised backward by target: matrix waves produce the d_aggr tiles ahead of the vector waves "
                                                "(csrc/triplet_ws_b1.hip); reads xw (gather), d_out, a_ij, stats, edge records"},
        "k_triplet_bwd_src_ws+dx": {"bound": "hbm", "bytes": b2 + f * N * C + img(HC + 8, C), "flops": 2 * N * (HC + 8) * C,
                                    "note": "warp-specialised backward by source with the d_x GEMM as the consumers' product (csrc/triplet_ws.hip)"},
        "k_triplet_bwd_src": {"bound": "hbm", "bytes": b2},
        "k_ts_gemm<4, 12, 4>": {"bound": "mfma", "flops": 2 * N * (HC + 8) * C, "bytes": f * N * (HC + 8 + C) + img(HC + 8, C),
                                "note": "d_x = [d_xw | d_a] @ Wcat^T as its own launch (general graphs beyond GLAM_FUSE_MAX_NODES)"},
        "k_reduce_partials": {"bound": "latency", "bytes": 0},
        # (the label glam_prof_* reports is the launch expression: the template argument is part of it)
        "k_wgrad<false>": {"bound": "mfma", "flops": 2 * N * (HC + 1) * C + 2 * N * (HC + 8) * C,
                           "bytes": f * N * (HC + C) + f * N * (HC + 8 + C),
                           "note": "both weight-gradient products ([aggr | 1]^T d_out, [d_xw | d_a]^T x) in one launch"},
        "k_wgrad_x3<false>": {"bound": "hbm", "flops": 2 * N * (HC + 1) * C + 2 * N * (HC + 8) * C,
                              "bytes": f * N * (HC + C) + f * N * (HC + 8 + C),
                              "note": "the same two products warp-specialised on the bf16 matrix cores in 3 x bf16 form (csrc/wgrad_x3.hip): "
                                      "memory-bound by construction, so it is priced against the HBM roofline"},
        "k_param_grads": {"bound": "latency", "bytes": 0},
    }


def rate(model, us):
    """roofline entry of one kernel from its model row and average duration."""
    out = {"avg_us": us, "bound": model["bound"]}
    if "bytes" in model and model["bytes"]:
        gbs = model["bytes"] / (us * 1e-6) / 1e9
        out.update(algorithmic_bytes=model["bytes"], achieved_GBs=gbs, frac_hbm_peak=gbs / HBM_PEAK_GBS,
                   frac_hbm_achievable=gbs / HBM_ACHIEVABLE_GBS)
    if "flops" in model:
        tf = model["flops"] / (us * 1e-6) / 1e12
        out.update(flops=model["flops"], achieved_TFLOPs=tf, frac_mfma_f32_peak=tf / MFMA_F32_PEAK_TFLOPS)
    return out


def committed_traffic(kernel_label, batch):
    """HBM bytes per launch of `kernel_label` from the committed rocprofv3 --pmc passes of this command (profiles/r*_hbm_traffic_pmc_b<B>.json:
    FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate passes; tools/rocpd_traffic.py).  PMC counters cannot be collected inside
    an unprofiled run: the figure quoted is the newest committed one, with its file name; None when there is none for this kernel."""
    import glob
    import re
    stem = re.sub(r"\+.*", "", kernel_label)               # "k_triplet_fwd_ws+update" -> "k_triplet_fwd_ws"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic_pmc_b{batch}.json")), reverse=True):
        try:
            rows = json.load(open(path))
        except Exception:       # noqa: BLE001
            continue
        rows = rows.get("kernels", rows) if isinstance(rows, dict) else rows
        items = rows.items() if isinstance(rows, dict) else ((r.get("kernel", ""), r) for r in rows)
        for name, r in items:
            if isinstance(r, dict) and stem + "<" in name.replace("glam::", "") + "<":
                tot = r.get("traffic_bytes") or r.get("hbm_bytes") or r.get("total_bytes")
                if tot:
                    return float(tot), os.path.relpath(path, ROOT)
    return None, None


def profile_step(body, reps, warm=3):
    """Average per-dispatch duration of every kernel `body` launches: {label: (avg_us, launches per body call)}."""
    from glam_amd import _lib
    for _ in range(warm):
        body()
    torch.cuda.synchronize()
    with _lib.kernel_timer(capacity=64 * reps) as kt:
        for _ in range(reps):
            body()
    torch.cuda.synchronize()
    acc = {}
    for name, grid, us in kt.records():
        a = acc.setdefault(name, [0.0, 0, grid])
        a[0] += us
        a[1] += 1
    return {k: {"avg_us": v[0] / v[1], "launches_per_step": v[1] / reps, "grid": v[2]} for k, v in acc.items()}


def time_isolated_aggregate(conv, batch, x, reps):
    """The aggregate kernels on their own (glam_triplet_fwd / glam_triplet_bwd: no fused GEMM epilogues)."""
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

    ell = gi.ell()      # index records of the software-pipelined forward (molecules: in-degree <= 4)
    onehot = int(ops.rows_are_one_hot(ea))

    def body():
        lib.glam_triplet_fwd(p(xw), p(a_ij), p(ea), p(We), p(M), p(gi.rowptr), p(gi.src), p(gi.eid), N, E, H, Cp, Dp, 1,
                             0.2, p(aggr), p(stats), st())
        if ell is not None:
            lib.glam_triplet_fwd_ell(p(xw), p(a_ij), p(ea), p(We), p(M), p(ell[0]), p(ell[1]), N, E, H, Cp, Dp, 0.2, onehot, p(aggr),
                                     p(stats), 0, st())
        lib.glam_triplet_bwd(p(xw), p(a_ij), p(ea), p(We), p(M), p(aggr), p(stats), p(d_aggr), p(gi.rowptr), p(gi.src),
                             p(gi.eid), p(colptr), p(dst), p(eid_t), N, E, H, Cp, Dp, 1, 0.2, p(d_xw), p(d_a), p(d_we),
                             p(d_M), None, p(ws), ws.numel(), st())
    return profile_step(body, reps)


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


def spawn_ranks(n):
    """``python bench.py --gpus N`` run as written: start the N ranks ourselves.  This process has made no GPU call yet
    (``device_count`` does not initialise the runtime) and never will: it only waits for its children — fresh interpreters,
    each of which binds ONE device — and returns the worst exit code.  Rank 0 inherits our stdout for the JSON line."""
    share = os.environ.get("GLAM_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < n and not share:
        print(f"bench.py: --gpus {n} but this node exposes {have} GPU(s) (GLAM_BENCH_SHARE_GPU=1 runs a functional check of the "
              f"multi-rank path on one device over gloo)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10000)     # ~1 s of replayed steps: long enough for an SMI sampler to see the GPU busy
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1024, help="molecules per GPU per step")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--eager-allreduce", action="store_true", help="keep the gradient all-reduce outside the captured graph")
    ap.add_argument("--steps-per-graph", type=int, default=16,
                    help="consecutive steps captured into one hipGraph launch (each graph launch carries a ~7 us bubble on this stack: "
                         "98.0 / 94.5 / 92.6 / 91.7 / 91.3 us per step at 1 / 2 / 4 / 8 / 16 steps per launch, tools/exp_multistep_graph.py)")
    ap.add_argument("--preheat-ms", type=float, default=50.0, help="untimed graph replays before the warm-up steps until the device clocks are back up (0 = none)")
    ap.add_argument("--stage-per-step", action="store_true", help="keep the parameter re-layout launch (k_stage_params) inside every step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget (0 = skip)")
    ap.add_argument("--large-batch", type=int, default=16384, help="extra roofline point beyond the LLC (0 = skip)")
    ap.add_argument("--prof-reps", type=int, default=30, help="eager profiled steps behind roofline_kernels")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    legs = {"import_torch": _T_TORCH - _T_START}      # wall_s per leg of this command (rank 0's clock)
    t_leg = [time.perf_counter()]

    def leg(name):
        now = time.perf_counter()
        legs[name] = legs.get(name, 0.0) + now - t_leg[0]
        t_leg[0] = now

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}", file=sys.stderr)
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
    # one route for everything this script issues: the Python autograd node (what a hipGraph capture records).  Eagerly issued steps would
    # otherwise go through the C++ node of the torch extension (same numbers, but the general kernels instead of the ELL ones), and
    # roofline_kernels — taken from eager executions of the captured function — would describe kernels the timed replay does not launch
    ops.USE_TORCH_EXT = False
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
    leg("setup (process group, synthetic batch, model to device)")
    params = list(conv.parameters())
    n_param = sum(p.numel() for p in params)
    flat = torch.zeros(n_param, device=dev)                    # gradient bucket when the grads are not one already
    live = {}

    # configs[1]'s step is forward + backward of the layer: no optimizer inside it, so the parameters stand still and their staged
    # images (GEMM weight images, W_edge, M, bias: k_stage_params) are built once, before the timed region (ops.cached_staging: they
    # are rebuilt whenever a parameter's address or version counter changes).  In a training step the re-layout belongs to the launch
    # that writes the parameters; --stage-per-step restores the launch inside every step.
    def compute():
        with (contextlib.nullcontext() if args.stage_per_step else ops.cached_staging()):
            return compute_inner()

    def compute_inner():
        out = conv(x, batch.edge_index, batch.edge_attr)
        grads = torch.autograd.grad(out, params + [x], grad_outputs=cot)   # loss = <out, cot>
        bucket = flat_view(grads[:-1])          # the fused layer hands its 5 parameter gradients back as one buffer
        if bucket is None:
            bucket = torch.cat([g_.reshape(-1) for g_ in grads[:-1]], out=flat)
        live["bucket"], live["d_x"] = bucket, grads[-1]
        return bucket

    def compute_and_reduce():
        dist.all_reduce(compute())              # ONE bucket, ONE RCCL call per step

    # warm-up on a side stream (stages the CSR + its transpose, which sync once per new batch; creates the RCCL communicator)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            compute_and_reduce() if world > 1 else compute()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    # ---- capture: the whole step, collective included, is ONE graph launch per rank ----
    graph, ar_in_graph = None, False
    if not args.no_graph:
        want_ar = world > 1 and not share and not args.eager_allreduce
        if want_ar:
            ok = 1
            try:
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_):
                    compute_and_reduce()
                graph, ar_in_graph = g_, True
            except Exception as exc:            # noqa: BLE001 - a runtime that cannot capture the collective: fall back, loudly
                print(f"bench.py[rank {rank}]: capturing the all-reduce failed ({type(exc).__name__}: {exc}); "
                      f"replaying the compute graph and issuing the all-reduce eagerly", file=sys.stderr)
                ok = 0
                torch.cuda.synchronize()
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank takes the same route
            if int(flag.item()) == 0:
                graph, ar_in_graph = None, False
        if graph is None:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                compute()

    # ---- several consecutive steps per graph launch: the same kernels, the same work per step, fewer launch bubbles.  Only when the
    #      whole step (collective included) is inside the graph; an eager all-reduce between steps keeps one step per launch ----
    S = max(1, args.steps_per_graph)
    if args.steps < 2 * S:
        S = max(1, args.steps)           # a short run (the driver's --steps 20) is ONE launch shape: a single graph of exactly `steps` steps
    graph_multi = None
    if graph is not None and S > 1 and (world == 1 or ar_in_graph):
        ok = 1
        try:
            gm = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gm):
                for _ in range(S):
                    compute_and_reduce() if world > 1 else compute()
            graph_multi = gm
        except Exception as exc:                # noqa: BLE001
            print(f"bench.py[rank {rank}]: capturing {S} steps per graph failed ({type(exc).__name__}: {exc}); one step per launch", file=sys.stderr)
            ok = 0
            torch.cuda.synchronize()
        if world > 1:
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                graph_multi = None

    def step():
        if graph is not None:
            graph.replay()
            if world > 1 and not ar_in_graph:
                dist.all_reduce(live["bucket"])
        elif world > 1:
            compute_and_reduce()
        else:
            compute()

    def run_steps(k):
        """Exactly k steps: whole multi-step graph launches first, single-step launches for the remainder."""
        if graph_multi is not None:
            for _ in range(k // S):
                graph_multi.replay()
            k -= k // S * S
        for _ in range(k):
            step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    leg("eager warm-up + hipGraph capture")
    # every graph object the timed region replays is replayed once beforehand (untimed, on top of the W warm-up steps): the first
    # replay of a hipGraph uploads it to the device, which a 20-step run would otherwise count as step time (87.6 vs 82.3 us per step)
    if graph_multi is not None and args.steps >= S:
        graph_multi.replay()
    if graph is not None and (graph_multi is None or args.steps % S):
        graph.replay()
        if world > 1 and not ar_in_graph:
            dist.all_reduce(live["bucket"])
    # device pre-heat (untimed, reported as config.preheat_steps): the GPU's clocks drop while the host synthesises the batch and
    # captures the graphs, and take tens of milliseconds of load to come back — a --steps 20 --warmup 5 run measured 81.8 us per step
    # against 77.6 with the same 20 steps behind 50 ms of replays (and 76.6 in the 10 000-step run).  The timed region stays exactly
    # `steps` steps behind `warmup` warm-up steps.
    preheat_steps = 0
    if args.preheat_ms > 0 and (graph is not None or graph_multi is not None):
        unit = S if graph_multi is not None else 1
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        run_steps(unit)
        torch.cuda.synchronize()
        dur = max(time.perf_counter() - t_h, 1e-6)
        if world > 1:       # every rank must replay the SAME number of times: the steps carry collectives
            t = torch.tensor([dur], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dur = float(t.item())
        n_rep = int(min(max(args.preheat_ms * 1e-3 / dur, 0.0), 5000.0))
        for _ in range(n_rep):
            run_steps(unit)
        torch.cuda.synchronize()
        preheat_steps = (1 + n_rep) * unit
    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    leg("warm-up steps + timed steps")

    # ---- the other staging mode, timed the same way: with the default (cached images) the headline step starts at the node
    #      GEMM; a training step re-lays the parameters out after every optimizer write, i.e. k_stage_params inside every step ----
    other = None
    if graph is not None and world == 1 and not args.stage_per_step:
        def compute_staged():
            return compute_inner()
        gs = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gs):
            for _ in range(S):
                compute_staged()
        reps = max(1, args.steps // S)
        gs.replay()
        for _ in range(max(1, args.warmup // S)):
            gs.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            gs.replay()
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        other = {"staging": "per_step (k_stage_params inside every step)", "ms_per_step": dts / (reps * S) * 1e3,
                 "value": B * reps * S / dts, "steps": reps * S}
        del gs
        leg("second timed region (staging per step)")

    launch = "eager" if graph is None else ("hipGraph replay (graphs uploaded by one untimed replay)" + (f", {S} steps per graph launch" if graph_multi is not None else "") +
                                            (" (all-reduce captured in the graph)" if ar_in_graph else
                                             (" + eager all-reduce" if world > 1 else "")))
    result = {
        "metric": "molecules/sec fwd+bwd on ESOL-shaped batches", "value": value, "unit": "molecules/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"ESOL-shaped batch={B}/GPU (N={N} atoms, E={E} directed bonds), single "
                               f"TripletMessage({C},{De},heads={H}) layer fwd+bwd, fp32",
                   "launch": launch, "preheat_steps": preheat_steps,
                   "staging": "per_step (k_stage_params inside every step)" if args.stage_per_step else
                              "cached (parameter images built once before the timed region: configs[1]'s step has no optimizer, the parameters "
                              "stand still; rounds 1-2 timed per_step, round 3 cached)",
                   "parallelism": f"dp{world}", "global_batch": B * world,
                   "collective": None if world == 1 else ("gloo (GLAM_BENCH_SHARE_GPU functional check)" if share else
                                                          f"RCCL all-reduce of one {n_param}-float bucket, {world} ranks")},
    }

    if other is not None:
        result["staging_per_step"] = other
    if rank == 0:
        # ---- roofline: per-dispatch durations of the kernels the timed step launches (same process, same stream, the function
        #      the graph captured, issued eagerly so that every launch can carry its own begin / end events) ----
        model = step_kernel_model(N, E, H, C, De)
        prof = profile_step(compute, args.prof_reps)
        kernels = {}
        for name, rec in sorted(prof.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_per_step"]):
            row = dict(rec)
            m = model.get(name)
            if m is not None:
                row.update(rate(m, rec["avg_us"]))
                if "note" in m:
                    row["note"] = m["note"]
            kernels[name] = row
        step_kernel_us = sum(r["avg_us"] * r["launches_per_step"] for r in prof.values())
        dom = next((n for n in ("k_triplet_fwd_ws+update", "k_triplet_fwd+update") if n in kernels), None)
        if dom is not None:
            k = kernels[dom]
            result["roofline"] = {"kernel": dom + " (what the step launches: gather + segment softmax + scatter-add + aggr @ W_scale + bias "
                                            "in one launch)",
                                  "bound": "hbm", "achieved": k["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": k["frac_hbm_peak"], "frac_of_achievable_6290": k["frac_hbm_achievable"],
                                  "traffic": None, "algorithmic_bytes": k["algorithmic_bytes"], "avg_launch_us": k["avg_us"],
                                  "grid": k["grid"], "workload": f"B={B}"}
            tr, src = committed_traffic(dom, B)
            result["roofline"]["traffic"] = tr
            result["roofline"]["traffic_note"] = (f"bytes per launch from the committed PMC passes of this command ({src}: FETCH_SIZE x 2 + WRITE_SIZE, "
                                                  "separate rocprofv3 --pmc runs); counters cannot be read inside an unprofiled run") if tr else \
                "PMC traffic is not collectable inside an unprofiled run and no committed pass names this kernel: see profiles/ and DESIGN.md §4"
            # the scatter-aggregate forward is the kernel BASELINE.json's metric names; by time the largest launch of the step may be
            # another one (the backward-by-target kernel since it absorbed the d_aggr GEMM): name it and its own fraction
            big = next(iter(kernels))
            result["roofline"]["largest_launch_of_the_step"] = {
                "kernel": big, "avg_launch_us": kernels[big]["avg_us"], "frac_hbm_peak": kernels[big].get("frac_hbm_peak"),
                "frac_mfma_f32_peak": kernels[big].get("frac_mfma_f32_peak")}
        leg("roofline_kernels (eager profiled steps)")
        result["roofline_kernels"] = {"source": f"glam_prof_* per-dispatch timestamps, {args.prof_reps} eager executions of the captured step function "
                                                "after the timed region", "sum_kernel_us_per_step": step_kernel_us, "kernels": kernels}
        # the inference forward (what the evaluation passes of the reference's trainer launch: no aggr / stats store), same batch
        def infer():
            with torch.no_grad(), ops.cached_staging():
                conv(x.detach(), batch.edge_index, batch.edge_attr)
        pi = profile_step(infer, args.prof_reps)
        result["roofline_inference"] = {n: dict(r, **(rate(model[n], r["avg_us"]) if n in model else {})) for n, r in pi.items()}
        leg("roofline_inference")
        iso = time_isolated_aggregate(conv, batch, x.detach(), args.prof_reps)
        result["roofline_isolated"] = {n: dict(r, **rate(model[n], r["avg_us"])) for n, r in iso.items() if n in model}
        leg("roofline_isolated")
        if args.large_batch and world == 1:
            big_cpu = synth_batch(args.large_batch, seed=7)
            big = big_cpu.to(dev)
            leg("roofline_large: synthetic batch")
            Nb, Eb = big.x.size(0), big.edge_index.size(1)
            xb = torch.randn(Nb, C, device=dev).requires_grad_(True)
            cb = torch.randn(Nb, C, device=dev)

            def compute_big():
                with (contextlib.nullcontext() if args.stage_per_step else ops.cached_staging()):
                    out = conv(xb, big.edge_index, big.edge_attr)
                    live["big"] = torch.autograd.grad(out, params + [xb], grad_outputs=cb)

            mb = step_kernel_model(Nb, Eb, H, C, De)
            # 40 untimed steps (~30 ms) first: the device idled while the batch was synthesised on the host, and its clocks take longer than
            # three steps to come back (the same kernels read 20 % slower with warm = 3 than in a dedicated --batch 16384 run)
            pb = profile_step(compute_big, max(5, args.prof_reps // 3), warm=40)
            ib = time_isolated_aggregate(conv, big, xb.detach(), max(5, args.prof_reps // 3))
            rl = {"workload": f"B={args.large_batch} (N={Nb}, E={Eb}: every [N,180] tensor is {Nb * 720 / 2 ** 20:.0f} MiB, beyond the 256 MiB LLC)",
                  "step_kernels": {n: dict(r, **(rate(mb[n], r["avg_us"]) if n in mb else {})) for n, r in pb.items()},
                  "isolated": {n: dict(r, **rate(mb[n], r["avg_us"])) for n, r in ib.items() if n in mb}}
            # frac = the fused forward scatter-aggregate kernel THE STEP LAUNCHES at this size (its update GEMM included, as at B = 1024);
            # the aggregate kernels on their own stay under "isolated"
            domb = next((n for n in ("k_triplet_fwd_ws+update", "k_triplet_fwd+update") if n in rl["step_kernels"]), None)
            if domb is not None:
                ki = rl["step_kernels"][domb]
                rl.update(kernel=domb + " (gather + segment softmax + scatter-add + update GEMM: what the step launches)", achieved=ki["achieved_GBs"],
                          frac=ki["frac_hbm_peak"], frac_of_achievable_6290=ki["frac_hbm_achievable"], avg_launch_us=ki["avg_us"],
                          algorithmic_bytes=ki["algorithmic_bytes"], bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s")
                trb, srcb = committed_traffic(domb, args.large_batch)
                rl["traffic"], rl["traffic_source"] = trb, srcb
            rl["sum_kernel_us_per_step"] = sum(r["avg_us"] * r["launches_per_step"] for r in pb.values())
            result["roofline_large"] = rl
            del big, xb, cb
            live.pop("big", None)
            leg("roofline_large")
        if args.cpu_seconds > 0 and world == 1:     # rank 0 at N = 1 only: the other ranks of a multi-GPU run would sit in the barrier
            result["cpu_baseline"] = cpu_baseline(batch_cpu, conv_cpu, x_cpu, cot_cpu, args.cpu_seconds)
            result["gpu_over_cpu"] = value / world / result["cpu_baseline"]["value"]
            leg("cpu_baseline")
        result["wall_s"] = dict({k: round(v, 2) for k, v in legs.items()}, total=round(time.perf_counter() - _T_START, 2))
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
