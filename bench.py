#!/usr/bin/env python3
"""bench.py -- graphs/sec of the DeSCo hot path (29-query neighborhood + gossip inference).

    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks when N > 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path (neighborhood counting -> apply_neighborhood_count ->
gossip propagation -> per-graph aggregation, main.py:296-302, 417-423 of the reference) over the
rank's resident shard: a COX2-shaped synthetic dataset (467 graphs, BASELINE.json configs[1])
replicated ``--replicas`` times so that one pass saturates the GPU.  Inputs (CSR blocks, weights)
are resident in HBM before the timed region.  Data parallel over graphs, no data-path collective.
``--scaling weak`` (default): every rank owns the same number of graphs; ``--scaling strong``: ONE
dataset (workload x replicas) is cut into cost-balanced graph ranges (distributed.shard_graphs)
and rank 0 assembles the per-graph counts of all ranks (InferencePipeline.gather).

Without a torchrun environment, ``--gpus N`` (N > 1) starts N worker processes itself BEFORE
anything touches the GPU (desco_amd.distributed.launch: fresh children, never an exec), each
joins the RCCL group and asserts world_size == N.

Prints ONE JSON line on rank 0 with the driver's contract plus
  "roofline":     dominant kernel, achieved vs peak from HIP events recorded live in the timed
                  region; "gather" = the fused SHMP layer kernel (the north_star's gather) priced
                  against HBM, with its PMC traffic
  "cpu_baseline": the CPU oracle (reference-form torch port) on PRE-BUILT batches, timed with
                  1 thread and with all physical cores on a bounded sample (N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s, ~6.3 achievable)
PEAK_F32_MFMA_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak (same guide)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak (same guide)
# fp32-accurate kernels on the bf16 pipe ("bf16x6"): every algorithmic fp32 multiply-add is six bf16
# MFMA products, so the speed of light of the ALGORITHM is the bf16 peak / 6 in fp32-equivalent flops
PEAK_X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
# ... and on the fp16 pipe in THREE products ("f16x3", round 4: hi/lo fp16 split with power-of-two scales; the fp16
# MFMA runs at the bf16 rate): the algorithm's speed of light is the fp16 peak / 3
PEAK_X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0


def gather_kernel() -> str:
    """Profiler key of the count-row launches of the fused SHMP layer (the gather-heavy kernel)."""
    from desco_amd import gnn_model, ops
    return ops.shmp_kernel_name(3, 2, True, gnn_model.SHMP_F16X3 and ops.pool_tile_rows() == 16)


def gather_mfma_peak() -> float:
    from desco_amd import gnn_model, ops
    return PEAK_X3_TFLOPS if gnn_model.SHMP_F16X3 and ops.pool_tile_rows() == 16 else PEAK_X6_TFLOPS



def mfma_peak(kernel: str):
    """(peak TFLOP/s in algorithmic fp32 flops, pipe) for an MFMA-bound kernel, else None."""
    if kernel == "gemm_f32_kernel" or kernel.endswith(",f32>"):
        return PEAK_F32_MFMA_TFLOPS, "v_mfma_f32_32x32x2_f32"
    if kernel in ("gemm_f16x3_kernel", "gossip_fused_f16_kernel", "post_tail_kernel") or ",f16x3" in kernel:
        return PEAK_X3_TFLOPS, "fp16 MFMA x 3 products (f16x3, fp32-accurate)"
    if kernel in ("gemm_split_kernel", "gossip_fused_kernel") or kernel.endswith(",x6>") or \
            kernel.startswith("shmp_layer16_kernel<"):
        return PEAK_X6_TFLOPS, "bf16 MFMA x 6 products (bf16x6, fp32-accurate)"
    return None


def build_models(device, seed=0, gains=(1.3, 1.4)):
    """Random-init weights of the reference architecture (no checkpoint is reachable offline):
    default nn.Linear init, matrices widened by ``gains`` (neighborhood, gossip model) so that 8 relu
    layers keep O(1), finite activations: (1.3, 1.4) for molecule-sized neighborhoods, (0.8, 1.2) for the
    dense shapes (rows there sum over many more neighbours; 2**logit must stay finite)."""
    import torch
    from desco_amd.lightning_model import GossipCountingModel, NeighborhoodCountingModel
    na = argparse.Namespace(layer_num=8, conv_type="SAGE", use_hetero=True, dropout=0.0, depth=4,
                            lr=1e-4, weight_decay=0.0, use_tconv=True, hidden_dim=64, input_dim=1,
                            batch_size=512)
    # (gossip dropout 0.01 = the reference default, config.py:316; it acts in training mode only)
    ga = argparse.Namespace(layer_num=2, conv_type="GOSSIP", use_hetero=False, dropout=0.01,
                            lr=1e-3, weight_decay=0.0, hidden_dim=64, batch_size=256)
    torch.manual_seed(seed)
    nm = NeighborhoodCountingModel(1, 64, na).to_hetero_old(True, True)
    gm = GossipCountingModel(1, 64, ga, emb_channels=64, input_pattern_emb=True)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m, gain in ((nm, gains[0]), (gm, gains[1])):
            for p in m.parameters():
                if p.dim() == 2:
                    p.mul_(gain)
                else:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
    return nm.to(device), gm.to(device)


def attainable_mfma():
    """Bare bf16 MFMA loops on random data on THIS device (tools/micro/mfma_peak, built by
    __graft_entry__.build()): what the matrix pipe sustains once the chip has lowered its clock under
    load -- the datasheet 2.5 PFLOP/s assumes 2.4 GHz.  {shape: TFLOP/s} or None if the tool is absent."""
    exe = os.path.join(ROOT, "tools", "micro", "mfma_peak")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    res = {}
    for line in out.splitlines():
        if "TFLOP/s" in line and line.startswith("v_mfma"):
            shape = line.split(",")[0].strip()
            tf = float(line.split(":")[1].split("TFLOP/s")[0])
            clk = float(line.split("clock")[1].split("MHz")[0])
            if tf > res.get(shape, {"TFLOPs": 0})["TFLOPs"]:
                res[shape] = {"TFLOPs": tf, "in_kernel_clock_MHz": clk}
    return res or None


def host_cpu_info():
    """(model string, physical cores, logical cores) from lscpu (fallback: os.cpu_count)."""
    logical = os.cpu_count() or 1
    model, phys = "unknown", logical
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {}
        for line in txt.splitlines():
            if ":" in line:
                k, v = line.split(":", 1)
                kv[k.strip()] = v.strip()
        model = kv.get("Model name", model)
        cps, soc = int(kv.get("Core(s) per socket", "0")), int(kv.get("Socket(s)", "0"))
        if cps > 0 and soc > 0:
            phys = cps * soc
    except Exception:      # lscpu missing: keep the fallback
        pass
    try:                   # cores this process may actually use (cgroup / affinity limits)
        phys = max(1, min(phys, len(os.sched_getaffinity(0))))
    except AttributeError:
        pass
    return model, phys, logical


def cpu_baseline(nm, gm, graphs_host, queries, target_seconds=16.0, workers=None):
    """The CPU oracle in the reference's form on a bounded sample of the same workload, on
    PRE-BUILT batches (canonical partition, triangle split and collate excluded, as on the GPU
    side -- SURVEY 8d), timed with k = 1 thread and with k = all physical cores."""
    import torch
    from oracle import model as OM
    sd_n = {k: v.detach().cpu().float() for k, v in nm.state_dict().items()}
    sd_g = {k: v.detach().cpu().float() for k, v in gm.state_dict().items()}
    model, phys, logical = host_cpu_info()
    # size the sample from a 2-graph probe with one thread
    torch.set_num_threads(1)
    probe = OM.prebuild_reference_inputs(graphs_host[:2], queries)
    t0 = time.perf_counter()
    OM.run_reference_prebuilt(sd_n, sd_g, probe, emulate_quirk=False)
    per_graph = (time.perf_counter() - t0) / 2
    n = int(max(2, min(4 * len(graphs_host), 0.5 * target_seconds / max(per_graph, 1e-6))))
    sample = [graphs_host[i % len(graphs_host)] for i in range(n)]     # cycles over the dataset
    t0 = time.perf_counter()
    pre = OM.prebuild_reference_inputs(sample, queries)
    t_prep = time.perf_counter() - t0
    runs, ref = {}, None
    for k in sorted({1, phys}):
        torch.set_num_threads(k)
        t0 = time.perf_counter()
        ref = OM.run_reference_prebuilt(sd_n, sd_g, pre, emulate_quirk=False)
        runs[k] = n / (time.perf_counter() - t0)
    best_k = max(runs, key=runs.get)
    torch.set_num_threads(max(1, min(8, phys)))
    value, cores, pp = runs[best_k], best_k, None
    if workers:
        # the honest all-core figure (SURVEY 8d "k = all physical host cores"): intra-op threading of thousands of
        # tiny torch ops only adds synchronisation, so P single-threaded processes take disjoint shards of a sample
        # sized for ~target_seconds / 2 of wall time at the 1-thread rate per process
        P = len(workers)
        # (drawn from the SAME graphs as the one-process sample: on a dataset ordered by size -- Syn_1827 -- cycling
        #  over all of it would hand every worker graphs a hundred times dearer than the ones the rate was sized on)
        n_pp = int(max(2 * P, min(64 * len(graphs_host), 0.5 * target_seconds * runs[1] * P)))
        sample_pp = [sample[i % len(sample)] for i in range(n_pp)]
        rate_wall, wall, slowest, fastest, rate = cpu_baseline_processes(workers, sd_n, sd_g, sample_pp, queries)
        pp = {"value": rate, "value_static_shards_wall_clock": rate_wall, "processes": P, "threads_per_process": 1,
              "graphs": n_pp, "wall_s": wall, "slowest_worker_s": slowest, "fastest_worker_s": fastest,
              "note": "value = sum of the workers' own rates (graphs of its shard / its time): the rate of the same "
                      "cores under dynamic work distribution; the wall-clock rate of the static equal shards is lower "
                      "by the spread between the workers"}
        if rate > value:
            value, cores = rate, P
    return {"value": value, "unit": "graphs/s", "cores": cores, "kind": "port",
            "value_process_parallel": None if pp is None else pp["value"], "process_parallel": pp,
            "value_1_thread": runs[1], "value_all_physical_cores": runs[phys], "physical_cores": phys,
            "logical_cores": logical, "cpu_model": model,
            "sample": f"{n} graphs cycling over the synthetic set; reference-form model (per-edge-type "
                      f"index_select/index_add_/Linear, 29-iteration loops, queries re-embedded per batch, "
                      f"batches 512/256, torch fp32) on PRE-BUILT collated batches: the oracle's Python "
                      f"canonical partition + triangle split + collate ({t_prep:.1f} s for the sample) is "
                      f"excluded, as the partition build is excluded from the GPU's timed region; "
                      f"value = best of one process with k=1 ({runs[1]:.1f}) / k={phys} threads ({runs[phys]:.1f}) and "
                      f"P single-threaded processes over disjoint graph shards "
                      f"({'not run' if pp is None else '%d processes: %.1f' % (pp['processes'], pp['value'])})"}, ref, n


# ---- process-parallel CPU baseline: P single-threaded worker processes over disjoint graph shards ------------------
def cpu_worker_main():
    """``bench.py --cpu-worker``: one single-threaded CPU-oracle worker.  Protocol on stdin / stdout:
    ``job <path> <index> <count>`` -> loads the job, pre-builds its shard's batches, answers ``ready``;
    ``go`` -> runs the reference-form model over the shard, answers ``done <seconds>``."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import torch
    torch.set_num_threads(1)
    from oracle import model as OM
    print("up", flush=True)               # imports done: from here on the process sleeps on stdin
    pre = sd_n = sd_g = None
    for line in sys.stdin:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "job":
            job = torch.load(tok[1], weights_only=False)
            i, cnt = int(tok[2]), int(tok[3])
            n = len(job["graphs"])
            lo, hi = (n * i) // cnt, (n * (i + 1)) // cnt
            sd_n, sd_g = job["sd_n"], job["sd_g"]
            pre = OM.prebuild_reference_inputs(job["graphs"][lo:hi], job["queries"]) if hi > lo else None
            print("ready", flush=True)
        elif tok[0] == "go":
            t0 = time.perf_counter()
            if pre is not None:
                OM.run_reference_prebuilt(sd_n, sd_g, pre, emulate_quirk=False)
            print(f"done {time.perf_counter() - t0:.6f}", flush=True)
        elif tok[0] == "quit":
            break
    return 0


def start_cpu_workers():
    """Start the worker processes of the process-parallel CPU baseline.  Called at the very top of main(), BEFORE this
    process touches the GPU (fresh children, one python each).  P = physical cores, bounded by free memory (a torch
    import is ~0.4 GB per process)."""
    _, phys, _ = host_cpu_info()
    P = phys
    try:
        with open("/proc/meminfo") as f:
            avail_kb = [int(l.split()[1]) for l in f if l.startswith("MemAvailable")][0]
        P = max(1, min(P, int((avail_kb / 1e6 - 24.0) / 0.6)))
    except Exception:
        P = min(P, 16)
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker"], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT) for _ in range(P)]
    return procs


def wait_cpu_workers_up(procs):
    """Block until every worker has finished importing (they then sleep on stdin): the timed GPU region must not share
    the host with 128 python start-ups."""
    for p in procs or []:
        assert p.stdout.readline().strip() == "up", "cpu worker failed to start"


def stop_cpu_workers(procs):
    for p in procs or []:
        try:
            p.stdin.write("quit\n")
            p.stdin.flush()
            p.stdin.close()
        except Exception:
            pass
    for p in procs or []:
        try:
            p.wait(timeout=20)
        except Exception:
            p.kill()


def cpu_baseline_processes(procs, sd_n, sd_g, sample, queries):
    """graphs/s of P single-threaded workers over disjoint contiguous shards of ``sample`` (wall clock from the common
    start signal to the last worker's answer; the pre-build of each shard's batches is outside, as everywhere)."""
    import tempfile
    import torch
    P = len(procs)
    fd, path = tempfile.mkstemp(suffix=".pt", prefix="desco_cpu_job_")
    os.close(fd)
    try:
        torch.save({"sd_n": sd_n, "sd_g": sd_g, "graphs": sample, "queries": queries}, path)
        for i, p in enumerate(procs):
            p.stdin.write(f"job {path} {i} {P}\n")
            p.stdin.flush()
        for p in procs:
            assert p.stdout.readline().strip() == "ready", "cpu worker failed to load its shard"
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        per = [float(p.stdout.readline().split()[1]) for p in procs]
        wall = time.perf_counter() - t0
    finally:
        os.unlink(path)
    # The shards are static and equal, the workers are not (core sharing, clocks: the slowest took 2x the fastest's time
    # in round 4): the wall-clock rate is what this static split delivers, the SUM of the workers' own rates is what
    # the same cores deliver with the work handed out dynamically -- the larger, and the one reported as the baseline.
    n = len(sample)
    shard = [(n * (i + 1)) // P - (n * i) // P for i in range(P)]
    rate_sum = sum(c / t for c, t in zip(shard, per) if t > 0)
    return n / wall, wall, max(per), min(per), rate_sum


def train_traffic(key, kernel):
    """HBM bytes per launch of a training leg's dominant kernel from the committed PMC passes of that leg
    (profiles/pmc_traffic.json, tools/profile_round.sh: FETCH_SIZE / WRITE_SIZE of `bench.py --train-only --train-leg ...`,
    averaged over every launch of the kernel in that run), or None."""
    try:
        ks = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["workloads"][key]["kernels"]
        # (the bf16 training products are the <WN, 1> instantiation of gemm_split_kernel: rocprofv3's name)
        e = ks.get(kernel) or ks[{"gemm_bf16_kernel": "gemm_split_kernel"}[kernel]]
        return e["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        return None


def train_leg(device, batch_size=512, stride=4, precision="fp32", epochs=1):
    """BASELINE configs 3 / 4 (Syn_1827 training): the neighborhood model's training step -- forward,
    backward (every op a C-ABI kernel, desco_amd.autograd), Adam -- on REAL-size batches: all 1 827
    Syn_1827-shaped graphs, the reference's batch of 512 neighborhoods (config.py:255; about 52 k rows per
    batch on average, 150 k+ for the largest graphs), labels = exact canonical counts of the 29 queries
    computed on the device (desco_canonical_counts_dev).  shuffle=False (main.py:195) fixes the batch
    stream; every ``stride``-th batch of the epoch is timed (the batches grow with the graph size, so a
    prefix is not representative), after one untimed pass over the same batches.  Reference:
    lightning_model.py:228-254, 160-173."""
    import torch
    from desco_amd import autograd as AG, ops, synthetic
    from desco_amd.batch import NeighborhoodBatch
    from desco_amd.data import STANDARD_QUERY_IDS, graph_atlas_plus
    from desco_amd.groundtruth import canonical_counts
    from desco_amd.partition import build_partition_device
    AG.set_precision(precision)
    nm, _ = build_models(device, gains=(0.8, 1.2))
    nm.set_queries(STANDARD_QUERY_IDS)
    gs = synthetic.WORKLOADS["syn_1827"]()
    t0 = time.perf_counter()
    part = build_partition_device(gs, 4, device)
    queries = [graph_atlas_plus(i) for i in STANDARD_QUERY_IDS]
    truth = canonical_counts(gs, queries, backend="auto").float()
    y_all = truth[torch.from_numpy(part.indicator)]
    t_prep = time.perf_counter() - t0
    starts = list(range(0, part.num_neigh, batch_size))[::stride]
    batches = [NeighborhoodBatch(part.slice(b0, b0 + batch_size), device, y=y_all[b0:b0 + batch_size])
               for b0 in starts]
    opt = nm.configure_optimizers()["optimizer"]

    def step(b):
        opt.zero_grad(set_to_none=True)
        loss = nm.training_step(b, 0)
        AG.backward(loss)
        opt.step()
        return loss

    first = last = None
    for b in batches:            # untimed pass: builds the backward indices of every batch, warms caches
        last = step(b).detach()  # (detached: a kept loss would keep its autograd graph -- and the AccumulateGrad
        first = last if first is None else first        # nodes of this stream -- alive into the capture below)
    torch.cuda.synchronize(device)
    ops.PROFILER.enabled = True
    ops.PROFILER.reset()
    t0 = time.perf_counter()
    for _ in range(epochs):      # (epochs > 1: tools/check_pass_is_native.sh --train, which varies the timed step count)
        for b in batches:
            last = step(b).detach()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / epochs
    ops.PROFILER.enabled = False
    first, last = float(first), float(last)
    summ = ops.PROFILER.summary()
    for v in summ.values():
        for k_ in ("ms", "calls", "launches", "flops", "bytes"):
            if k_ in v:
                v[k_] = v[k_] / epochs
    tot = sum(v["ms"] for v in summ.values())
    n = sum(b.num_graphs for b in batches)
    rows = sum(b.num_rows for b in batches)
    # the same steps replayed from hipGraphs (what Trainer(graph_capture=True) / main.py --graph_capture do:
    # shuffle=False makes every epoch the same batch stream, so a batch's whole step -- forward, backward,
    # Adam -- is captured once and replayed): the eager step is host-bound
    # (~216 launches + autograd bookkeeping around ~5 ms of kernels)
    # (desco_amd.optim.Adam keeps its state on the device and its step is one capturable launch: the same optimizer
    #  object goes on)
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    graphs = []
    with torch.cuda.stream(side):
        for b in batches[:2]:          # autograd's AccumulateGrad nodes must have been created on this stream
            opt.zero_grad(set_to_none=True)
            AG.backward(nm.train_forward(b, 0))
            opt.step()
        for b in batches:
            opt.zero_grad(set_to_none=True)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg, stream=side):
                AG.backward(nm.train_forward(b, 0))
                opt.step()
            graphs.append(cg)
        for cg in graphs[:4]:
            cg.replay()
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(epochs):
        for cg in graphs:
            cg.replay()
    torch.cuda.synchronize(device)
    dt_graph = (time.perf_counter() - t0) / epochs
    del graphs
    dt_eager, dt = dt, dt_graph
    name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
    peak = {"gemm_f32_kernel": PEAK_F32_MFMA_TFLOPS, "linear_bwd_w_kernel": PEAK_F32_MFMA_TFLOPS,
            "gemm_f32_multi_kernel": PEAK_F32_MFMA_TFLOPS, "linear_bwd_w_multi_kernel": PEAK_F32_MFMA_TFLOPS,
            "gemm_tn_partial_kernel": PEAK_F32_MFMA_TFLOPS, "gemm_bf16_kernel": PEAK_BF16_MFMA_TFLOPS,
            "gemm_split_kernel": PEAK_X6_TFLOPS}.get(name)
    if peak:
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        roof = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak}
    else:
        ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        roof = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS}
    roof.update({"launches": d["calls"], "avg_launch_ms": d["ms"] / d["calls"], "share_of_kernel_time": d["ms"] / tot,
                 "traffic": train_traffic(f"train_{precision}", name)})
    AG.set_precision("fp32")
    return {
        "metric": "neighborhoods/s (neighborhood-model training step: forward, backward, Adam)",
        "value": n / dt, "unit": "neighborhoods/s", "ms_per_step": 1e3 * dt / len(batches),
        "rows_per_s": rows / dt, "steps": len(batches), "rows_per_step": rows / len(batches),
        "max_rows_per_step": max(b.num_rows for b in batches), "kernel_ms_per_step": tot / len(batches),
        "launches_per_step": sum(v["launches"] for v in summ.values()) / len(batches),
        "launch_mode": "hipGraph replay per batch",
        "eager": {"value": n / dt_eager, "ms_per_step": 1e3 * dt_eager / len(batches),
                  "note": "same steps as eager launches with per-launch HIP events (host-bound)"},
        "dtype": "f32" if precision == "fp32" else "bf16 products, fp32 accumulate", "data": "synthetic",
        "loss_first_pass_first_batch": float(first), "loss_last_batch": float(last),
        "config": {"workload": f"Syn_1827-shaped synthetic, all {gs.num_graphs} graphs ({part.num_neigh} neighborhoods, "
                               f"{part.num_rows} rows), batch {batch_size} neighborhoods, every {stride}-th batch of the "
                               f"epoch, 29 queries, exact canonical-count labels, Adam (desco_adam_step_f32)",
                   "prep_s": round(t_prep, 2)},
        "roofline": roof,
        "kernels": {k: {"launches_per_step": round(v["launches"] / len(batches), 1),
                        "ms_per_step": round(v["ms"] / len(batches), 3)}
                    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:10]},
    }


def ddp_world1_child(stride=4, batch_size=512, epochs=2):
    """Child of ``ddp_world1_leg`` (a fresh process that has not touched the GPU; WORLD_SIZE=1,
    DESCO_FORCE_COLLECTIVES=1): the neighborhood training leg's steps through the DATA-PARALLEL code path against real
    RCCL -- init_process_group("nccl", device_id=...), parameter broadcast, the hook-driven eager step (asynchronous
    bucket all-reduces issued from autograd hooks on the training stream), then trainer.DDPReplay (graph A, bucket
    all-reduces, graph B).  One rank reduces with itself, so the numbers price the path, not the wire."""
    import torch
    from desco_amd import autograd as AG, distributed as D, synthetic
    from desco_amd.batch import NeighborhoodBatch
    from desco_amd.data import STANDARD_QUERY_IDS, graph_atlas_plus
    from desco_amd.groundtruth import canonical_counts
    from desco_amd.partition import build_partition_device
    from desco_amd.trainer import DDPReplay
    device = torch.device("cuda", 0)
    assert D.init_from_env(device, backend="nccl") and D.collectives_on()
    ones = torch.ones(1, device=device)
    D.all_reduce_(ones, "sum")
    backend = torch.distributed.get_backend()
    nm, _ = build_models(device, gains=(0.8, 1.2))
    nm.set_queries(STANDARD_QUERY_IDS)
    gs = synthetic.WORKLOADS["syn_1827"]()
    part = build_partition_device(gs, 4, device)
    truth = canonical_counts(gs, [graph_atlas_plus(i) for i in STANDARD_QUERY_IDS], backend="auto").float()
    y_all = truth[torch.from_numpy(part.indicator)]
    starts = list(range(0, part.num_neigh, batch_size))[::stride]
    batches = [NeighborhoodBatch(part.slice(b0, b0 + batch_size), device, y=y_all[b0:b0 + batch_size]) for b0 in starts]
    D.broadcast_params(nm)
    opt = nm.configure_optimizers()["optimizer"]
    buckets = D.GradBuckets(list(nm.parameters()), 4)
    ddp = DDPReplay(nm, opt, buckets, device)
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    losses = []
    with torch.cuda.stream(side):
        t_eager = None
        for ep in range(2):                       # hook-driven eager steps (the first one settles the bucket layout)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i, b in enumerate(batches):
                buckets.zero()
                loss = nm.train_forward(b, i)
                AG.backward(loss, 1.0)
                buckets.finish()
                opt.step()
                if ep == 0 and i == 0:
                    losses.append(float(loss.detach()))
            torch.cuda.synchronize(device)
            t_eager = time.perf_counter() - t0
        for k, b in enumerate(batches):           # capture (graph A per batch, graph B once) + first replay
            ddp.step(k, k, b, 1.0, stream=side)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(epochs):
            for k, b in enumerate(batches):
                ddp.step(k, k, b, 1.0, stream=side)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / epochs
    torch.cuda.current_stream(device).wait_stream(side)
    nm.invalidate_caches()
    with torch.no_grad():
        losses.append(float(nm.train_forward(batches[0], 0)))
    n = sum(b.num_graphs for b in batches)
    rec = {"value": n / dt, "unit": "neighborhoods/s", "ms_per_step": 1e3 * dt / len(batches), "steps": len(batches),
           "launch_mode": "hipGraph replay: graph A (zero, forward, backward, pack) + bucket all-reduces + graph B (Adam)",
           "eager_hooks": {"ms_per_step": 1e3 * t_eager / len(batches),
                           "note": "hook-driven asynchronous bucket all-reduces during backward, eager launches"},
           "collective": {"backend": backend, "world_size": D.world_size(), "ranks_seen": int(round(float(ones.item()))),
                          "buckets": len(buckets.buckets), "bucket_bytes": [int(4 * f.numel()) for f in buckets.buckets],
                          "device_id_bound": True},
           "loss_first_batch_before": losses[0], "loss_first_batch_after": losses[1]}
    buckets.close()
    D.barrier()
    torch.distributed.destroy_process_group()
    print(json.dumps({"ddp_world1_nccl": rec}))


def ddp_world1_leg(stride=4):
    """``train_syn_1827.ddp_world1_nccl``: start ``ddp_world1_child`` as a fresh process (RCCL wants its communicator
    made before anything else used the device in that process, and an exec from a process that initialised the GPU is
    not allowed on this pool: the child is spawned, never exec'd into)."""
    from desco_amd import distributed as D
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(D.free_port()), DESCO_FORCE_COLLECTIVES="1")
    env.pop("DESCO_SHARE_GPU", None)
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--ddp-child", "--train-stride", str(stride)],
                           capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"ddp_world1_nccl"')]
        if p.returncode != 0 or not lines:
            return {"status": "error", "returncode": p.returncode, "stderr_tail": p.stderr[-600:]}
        return json.loads(lines[-1])["ddp_world1_nccl"]
    except Exception as e:      # noqa: BLE001  (a failing side leg must not cost the measurement)
        return {"status": "error", "error": f"{type(e).__name__}: {e}"[:300]}


def train_gossip_leg(device, batch_graphs=256, epochs=2):
    """BASELINE config 4's second stage (Syn_1827 full training): the gossip model's training step -- forward, backward
    (every op a C-ABI kernel, desco_amd.autograd), Adam -- on all 1 827 Syn_1827-shaped graphs in the reference's batches
    of 256 graphs (config.py:319), node inputs = the exact canonical counts perturbed by 10 % (what a trained
    neighborhood stage hands over), labels = the exact counts; loss = sum log2(|pred - y| + 1)
    (lightning_model.py:585-608, 630-635).  One untimed epoch, then two timed epochs with per-launch HIP events."""
    import torch
    from desco_amd import autograd as AG, ops, synthetic
    from desco_amd.batch import GossipBatch
    from desco_amd.data import STANDARD_QUERY_IDS, graph_atlas_plus
    from desco_amd.groundtruth import canonical_counts
    nm, gm = build_models(device, gains=(0.8, 1.2))
    nm.set_queries(STANDARD_QUERY_IDS)
    gm.set_query_emb(nm.get_query_emb().detach())
    gs = synthetic.WORKLOADS["syn_1827"]()
    t0 = time.perf_counter()
    queries = [graph_atlas_plus(i) for i in STANDARD_QUERY_IDS]
    y = canonical_counts(gs, queries, backend="auto").float()
    g = torch.Generator().manual_seed(5)
    x = y * (1.0 + 0.1 * torch.randn(y.shape, generator=g)).clamp_min(0.0)
    t_prep = time.perf_counter() - t0
    batches, off = [], 0
    for g0 in range(0, gs.num_graphs, batch_graphs):
        sub = gs.subset(g0, min(g0 + batch_graphs, gs.num_graphs))
        batches.append(GossipBatch(sub, device, x=x[off:off + sub.num_nodes], y=y[off:off + sub.num_nodes]))
        off += sub.num_nodes
    opt = gm.configure_optimizers()["optimizer"]

    def step(b):
        opt.zero_grad(set_to_none=True)
        loss = gm.training_step(b, 0)
        AG.backward(loss)
        opt.step()
        return loss.detach()

    first = None
    for b in batches:
        l = step(b)
        first = l if first is None else first
    torch.cuda.synchronize(device)
    ops.PROFILER.enabled = True
    ops.PROFILER.reset()
    t0 = time.perf_counter()
    for _ in range(epochs):
        for b in batches:
            last = step(b)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    ops.PROFILER.enabled = False
    summ = ops.PROFILER.summary()
    tot = sum(v["ms"] for v in summ.values())
    steps = epochs * len(batches)
    nodes = epochs * sum(b.num_nodes for b in batches)
    dt_eager = dt
    # the same steps replayed from hipGraphs (Trainer(graph_capture=True)): one graph per batch, dropout included -- its
    # masks are functions of a (seed, step) pair in device memory that a captured launch advances
    gm.train()
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    graphs = []
    with torch.cuda.stream(side):
        for b in batches[:2]:          # autograd's AccumulateGrad nodes must have been created on this stream
            step(b)
        for b in batches:
            opt.zero_grad(set_to_none=True)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg, stream=side):
                AG.backward(gm.train_forward(b, 0))
                opt.step()
            graphs.append(cg)
        for cg in graphs[:2]:
            cg.replay()
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(epochs):
        for cg in graphs:
            cg.replay()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    del graphs
    name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
    peak = {"gemm_f32_kernel": PEAK_F32_MFMA_TFLOPS, "linear_bwd_w_kernel": PEAK_F32_MFMA_TFLOPS,
            "gemm_f32_multi_kernel": PEAK_F32_MFMA_TFLOPS, "linear_bwd_w_multi_kernel": PEAK_F32_MFMA_TFLOPS,
            "gemm_tn_partial_kernel": PEAK_F32_MFMA_TFLOPS}.get(name)
    if peak:
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        roof = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak}
    else:
        ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        roof = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS}
    roof.update({"launches": d["calls"], "avg_launch_ms": d["ms"] / d["calls"], "share_of_kernel_time": d["ms"] / tot,
                 "traffic": train_traffic("train_gossip", name)})
    return {
        "metric": "nodes/s (gossip-model training step: forward, backward, Adam; 29 queries per node)",
        "value": nodes / dt, "unit": "nodes/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
        "node_query_rows_per_s": 29.0 * nodes / dt, "kernel_ms_per_step": tot / steps,
        "launches_per_step": sum(v["launches"] for v in summ.values()) / steps, "launch_mode": "hipGraph replay per batch",
        "eager": {"value": nodes / dt_eager, "ms_per_step": 1e3 * dt_eager / steps,
                  "note": "same steps as eager launches with per-launch HIP events"},
        "dropout": float(gm.emb_model.gnn_core.dropout),
        "dtype": "f32", "data": "synthetic", "loss_first_batch": float(first), "loss_last_batch": float(last),
        "config": {"workload": f"Syn_1827-shaped synthetic, all {gs.num_graphs} graphs ({gs.num_nodes} nodes), batch "
                               f"{batch_graphs} graphs, 29 queries, inputs = exact canonical counts +-10 %, labels = exact "
                               f"counts, Adam (desco_adam_step_f32)", "prep_s": round(t_prep, 2)},
        "roofline": roof,
        "kernels": {k: {"launches_per_step": round(v["launches"] / steps, 1), "ms_per_step": round(v["ms"] / steps, 3)}
                    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]},
    }


def nccl_gradient_check(device):
    """One REAL neighborhood training step per rank on its (unequal) share of a union batch, the gradient buckets
    all-reduced asynchronously from the autograd hooks (desco_amd.distributed.GradBuckets), compared with the
    single-process gradient of the union batch.  Runs inside an initialised process group of any size; returns the
    record every rank agrees on.  Reference: main.py:242-255 (Lightning's ddp strategy)."""
    import torch
    from desco_amd import distributed as D
    from desco_amd import synthetic
    from desco_amd.batch import NeighborhoodBatch
    from desco_amd.data import STANDARD_QUERY_IDS
    from desco_amd.partition import build_partition
    rank, world = D.rank(), D.world_size()
    nm, _ = build_models(device, gains=(0.8, 1.2))
    nm.set_queries(STANDARD_QUERY_IDS)
    gs = synthetic.syn_1827_shaped(60)
    part = build_partition(gs, 4)
    B = min(part.num_neigh, 768)
    g = torch.Generator().manual_seed(4)
    y = torch.floor(torch.rand(B, len(STANDARD_QUERY_IDS), generator=g) ** 3 * 40)
    # unequal contiguous shares (count-weighted mean loss): cuts at i*B/world shifted by 37 where that fits
    cuts = [0] + [min(B, max(1, (i * B) // world + (37 if B // world > 74 else 0))) for i in range(1, world)] + [B]
    lo, hi = cuts[rank], cuts[rank + 1]
    mine = NeighborhoodBatch(part.slice(lo, hi), device, y=y[lo:hi])
    params = [p for p in nm.parameters() if p.requires_grad]
    bk = D.GradBuckets(params, 4)
    bk.zero()
    (nm.training_step(mine, 0) * ((hi - lo) / B)).backward()     # bucket all-reduces start from the hooks
    issued = bk._next
    bk.finish()
    got = [p.grad.detach().clone() for p in params]
    bk.close()
    for p in params:
        p.grad = None
    union = NeighborhoodBatch(part.slice(0, B), device, y=y)
    nm.training_step(union, 0).backward()
    worst = 0.0
    for gdist, p in zip(got, params):
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        den = float(ref.abs().max()) + 1e-12
        worst = max(worst, float((gdist - ref).abs().max()) / den)
    t = torch.tensor([worst], device=device)
    D.all_reduce_(t, "max")
    return {"status": "ok" if float(t) < 1e-4 else "FAILED", "world": world, "backend": torch.distributed.get_backend(),
            "buckets": len(bk.buckets), "buckets_issued_from_hooks": issued,
            "grad_bytes": sum(b.numel() for b in bk.buckets) * 4, "worst_rel_grad_diff_vs_union_batch": float(t)}


def nccl_selftest(args):
    """--selftest-nccl: the RCCL path without an 8-GPU node.  With >= 2 visible devices, two ranks (one per GPU, backend
    "nccl") run ``nccl_gradient_check``; prints one JSON line.  Skips cleanly (status "skipped") on a 1-device box.
    (With world > 1 and the nccl backend, the normal bench run performs the same check at start-up and reports it under
    "collective".)"""
    import torch
    from desco_amd import distributed as D
    ndev = torch.cuda.device_count()
    if "WORLD_SIZE" not in os.environ:
        if ndev < 2:
            print(json.dumps({"selftest": "nccl", "status": "skipped", "reason": f"{ndev} visible device(s); needs 2"}))
            return 0
        return D.launch([os.path.abspath(__file__), "--selftest-nccl"], 2, devices=[0, 1], timeout=600)
    rank, world, _ = D.env_world()
    device = D.local_device()
    D.init_from_env(device, backend="nccl")
    rec = nccl_gradient_check(device)
    if rank == 0:
        print(json.dumps(dict({"selftest": "nccl"}, **rec)))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
    return 0 if rec["status"] == "ok" else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cox2", choices=["cox2", "mutag", "syn_1827", "msrc_imdb"])
    ap.add_argument("--replicas", type=int, default=64, help="dataset replication factor per rank "
                    "(weak scaling) or of the one global dataset (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch HIP events")
    ap.add_argument("--no-x1", action="store_true", help="skip the small-dataset (x1) latency line")
    ap.add_argument("--no-attainable", action="store_true",
                    help="skip the bare-MFMA micro-benchmark next to the roofline (about 8 s)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short runs of the other workload shapes (Syn_1827, MSRC+IMDB)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the pass from a captured hipGraph (implies --no-profile)")
    ap.add_argument("--neigh-rows", type=int, default=48_000_000,
                    help="row budget of a neighborhood block (InferencePipeline max_neigh_rows)")
    ap.add_argument("--gossip-rows", type=int, default=48_000_000,
                    help="(node x query) row budget of a gossip block (InferencePipeline max_gossip_rows)")
    ap.add_argument("--chunks", type=int, default=None,
                    help="placement-independent mode (InferencePipeline chunks): N ranks reproduce the 1-rank "
                         "result bit for bit; a multiple of --gpus")
    ap.add_argument("--by-shape", action="store_true",
                    help="diagnostic: key GEMM launches by shape in the kernel table")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the training leg (Syn_1827-shaped neighborhood training steps, N=1 only)")
    ap.add_argument("--train-stride", type=int, default=4, help="training leg: time every k-th batch of the epoch")
    ap.add_argument("--train-precision", default="both", choices=["fp32", "bf16", "both"])
    ap.add_argument("--train-only", action="store_true",
                    help="run the training legs only and print their records as one JSON line (tools/check_pass_is_native.sh)")
    ap.add_argument("--train-leg", default="all", choices=["all", "fp32", "bf16", "gossip"],
                    help="--train-only: run one leg only (the PMC passes of tools/profile_round.sh take them one at a time)")
    ap.add_argument("--train-epochs", type=int, default=1,
                    help="timed passes over the training legs' batches (default 1; the gossip leg runs twice as many)")
    ap.add_argument("--selftest-nccl", action="store_true",
                    help="2-rank RCCL gradient all-reduce check (needs 2 visible GPUs; skips otherwise)")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--ddp-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        sys.exit(cpu_worker_main())
    if args.ddp_child:
        ddp_world1_child(stride=args.train_stride)
        return
    if args.selftest_nccl:
        sys.exit(nccl_selftest(args))

    from desco_amd import distributed as D      # (imports torch; no GPU call)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start the N ranks ourselves, before any GPU call in this process
        sys.exit(D.launch([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    # the process-parallel CPU baseline's workers are started NOW, before this process touches the GPU
    cpu_workers = None
    if args.gpus == 1 and not args.no_cpu_baseline and "WORLD_SIZE" not in os.environ:
        cpu_workers = start_cpu_workers()

    import torch
    rank, world, local_rank = D.env_world()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher must "
                         f"start exactly --gpus ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # DESCO_SHARE_GPU=1 (testing only): all ranks on device 0 with the gloo backend, to exercise the
    # multi-rank control flow on a 1-GPU box; real runs use one GPU per rank and RCCL ("nccl")
    share = D.share_gpu()
    device = D.local_device()
    D.init_from_env(device)
    assert D.world_size() == args.gpus, (D.world_size(), args.gpus)
    if args.train_only:
        precisions = ["fp32", "bf16"] if args.train_precision == "both" else [args.train_precision]
        if args.train_leg in ("fp32", "bf16"):
            precisions = [args.train_leg]
        rec = {}
        if args.train_leg != "gossip":
            rec["train_syn_1827"] = {p_: train_leg(device, stride=args.train_stride, precision=p_, epochs=args.train_epochs)
                                     for p_ in precisions}
            if args.train_leg == "all":
                torch.cuda.empty_cache()
                rec["train_syn_1827"]["ddp_world1_nccl"] = ddp_world1_leg(stride=args.train_stride)
        if args.train_leg in ("all", "gossip"):
            rec["train_gossip"] = train_gossip_leg(device, epochs=2 * args.train_epochs)
        print(json.dumps(rec))
        return
    # self-proof of the collective path: which backend is live and how many ranks it reaches
    collective = None
    if world > 1:
        ones = torch.ones(1, device=torch.device("cpu") if share else device)
        D.all_reduce_(ones, "sum")
        collective = {"backend": torch.distributed.get_backend(), "ranks_seen": int(round(float(ones.item()))),
                      "shared_gpu_test_mode": bool(share)}
        assert collective["ranks_seen"] == world, collective
        # the gradient all-reduce check on the live backend: always for nccl (RCCL, < 2 s); in the shared-GPU test mode
        # (gloo, host-staged) only when asked, since the test-suite times these runs
        if collective["backend"] == "nccl" or os.environ.get("DESCO_BENCH_GRAD_CHECK") == "1":
            # (a failing self-test must not cost the measurement: the inference pass below uses no collective but the
            #  barrier and the max over ranks)
            try:
                collective["grad_allreduce_selftest"] = nccl_gradient_check(device)
            except Exception as e:      # noqa: BLE001
                collective["grad_allreduce_selftest"] = {"status": "error", "error": f"{type(e).__name__}: {e}"[:300]}

    from desco_amd import ops, synthetic
    from desco_amd.data import STANDARD_QUERY_IDS
    from desco_amd.pipeline import InferencePipeline

    base = synthetic.WORKLOADS[args.workload]()
    graphs = base.replicate(args.replicas)
    # (molecule-sized neighborhoods: gains (1.3, 1.4); the dense shapes need narrower weights for finite 2**logit)
    nm, gm = build_models(device, gains=(1.3, 1.4) if args.workload in ("cox2", "mutag") else (0.8, 1.2))
    nm.set_queries(STANDARD_QUERY_IDS)
    strong = args.scaling == "strong"
    t0 = time.perf_counter()
    pipe = InferencePipeline(nm, gm, graphs, depth=4, device=device, max_neigh_rows=args.neigh_rows,
                             max_gossip_rows=args.gossip_rows,
                             rank=rank if strong else 0, world=world if strong else 1, chunks=args.chunks)
    t_build = time.perf_counter() - t0
    part = pipe.partition

    def sync():
        torch.cuda.synchronize(device)
        D.barrier()
        torch.cuda.synchronize(device)

    if args.graph:
        args.no_profile = True
        pipe.capture()
    wait_cpu_workers_up(cpu_workers)
    step = pipe.run_graph if args.graph else pipe.run
    for _ in range(args.warmup):
        out = step()
    sync()
    # Per-launch HIP events: ONE pass before the timed region brackets every launch (the kernel table of the line); the
    # timed region brackets the launches of the dominant kernel only -- what `roofline` needs -- because two events per
    # launch on all ~66 launches of a pass cost 0.5-1.5 % of the pass (eager 29.5 ms without, 29.7-30.1 with).
    full_pass = {}
    ops.PROFILER.by_shape = args.by_shape
    if not args.no_profile:
        ops.PROFILER.enabled = True
        ops.PROFILER.reset()
        out = step()
        sync()
        ops.PROFILER.enabled = False
        full_pass = ops.PROFILER.summary()
        ops.PROFILER.only = {max(full_pass.items(), key=lambda kv: kv[1]["ms"])[0]}
    ops.PROFILER.enabled = not args.no_profile
    ops.PROFILER.reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    ops.PROFILER.enabled = False
    ops.PROFILER.only = None
    comm_dev = torch.device("cpu") if share else device
    if world > 1:
        t = torch.tensor([elapsed], device=comm_dev, dtype=torch.float64)
        D.all_reduce_(t, "max")
        elapsed = float(t.item())
    # the only exchange of the inference path: per-graph counts to rank 0
    if strong:
        gathered = pipe.gather(out)
        if rank == 0:
            assert gathered["graph_gossip_count"].shape[0] == graphs.num_graphs
            assert torch.isfinite(gathered["graph_gossip_count"]).all()
    elif world > 1:
        g_all = D.gather_rows(out["graph_gossip_count"])
        if rank == 0:
            assert g_all.shape[0] == graphs.num_graphs * world and torch.isfinite(g_all).all()

    primary_summary = {}
    if not args.no_profile:
        # the table: the fully bracketed pass, scaled to the timed steps; the dominant kernel's entry: the timed region's own
        primary_summary = {k: {f: v[f] * args.steps for f in ("calls", "launches", "ms", "flops", "bytes")}
                           for k, v in full_pass.items()}
        primary_summary.update(ops.PROFILER.summary())
    # secondary workloads (every rank runs them; rank 0 reports): weak scaling, 3 timed steps
    secondary = {}
    if not args.no_secondary:
        # narrower random-init weights for the dense shapes, so that every output is finite and checked
        nm2, gm2 = build_models(device, gains=(0.8, 1.2))
        nm2.set_queries(STANDARD_QUERY_IDS)
        for wname, wrep in (("syn_1827", 2), ("msrc_imdb", 8)):
            if wname == args.workload:
                continue
            g2 = synthetic.WORKLOADS[wname]().replicate(wrep)
            p2 = InferencePipeline(nm2, gm2, g2, depth=4, device=device, max_neigh_rows=args.neigh_rows,
                                   max_gossip_rows=args.gossip_rows, rank=0, world=1)
            o2 = p2.run()
            sync()
            assert torch.isfinite(o2["graph_gossip_count"]).all() and torch.isfinite(o2["node_count"]).all(), wname
            del o2
            # (as in the primary run: one fully bracketed pass, then the timed passes with events around the dominant
            # kernel's launches only)
            ops.PROFILER.enabled = True
            ops.PROFILER.reset()
            p2.run()
            sync()
            ops.PROFILER.enabled = False
            full2 = ops.PROFILER.summary()
            ops.PROFILER.only = {max(full2.items(), key=lambda kv: kv[1]["ms"])[0], gather_kernel()}
            ops.PROFILER.enabled = True
            ops.PROFILER.reset()
            t0 = time.perf_counter()
            for _ in range(3):
                p2.run()
            sync()
            dt = time.perf_counter() - t0
            ops.PROFILER.enabled = False
            ops.PROFILER.only = None
            if world > 1:
                t = torch.tensor([dt], device=comm_dev, dtype=torch.float64)
                D.all_reduce_(t, "max")
                dt = float(t.item())
            summ2 = {k: {f: v[f] * 3 for f in ("calls", "launches", "ms", "flops", "bytes")} for k, v in full2.items()}
            summ2.update(ops.PROFILER.summary())
            tot2 = sum(d["ms"] for d in summ2.values())
            gk = summ2.get(gather_kernel())
            dom, dd = max(summ2.items(), key=lambda kv: kv[1]["ms"])
            entry = {"value": g2.num_graphs * world * 3 / dt, "unit": "graphs/s", "ms_per_step": 1e3 * dt / 3,
                     "outputs": "finite (checked); random-init weights with gains (0.8, 1.2)",
                     "graphs_per_gpu": g2.num_graphs, "neighborhood_rows_per_gpu": p2.partition.num_rows,
                     "neighborhood_directed_edges_per_gpu": p2.partition.num_edges,
                     "dominant_kernel": dom, "dominant_share_of_kernel_time": dd["ms"] / tot2}
            if gk and gk["ms"] > 0:
                gbs = gk["bytes"] / (gk["ms"] * 1e-3) / 1e9
                tfs = gk["flops"] / (gk["ms"] * 1e-3) / 1e12
                entry["gather"] = {"kernel": gather_kernel(), "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS,
                                   "mfma_TFLOPs": tfs, "mfma_frac_of_split_peak": tfs / gather_mfma_peak(),
                                   "share_of_kernel_time": gk["ms"] / tot2,
                                   "avg_launch_ms": gk["ms"] / gk["calls"]}
                # HBM bytes of these launches from the committed PMC passes of THIS shape, under the same validity
                # rules as the primary line (same launches per step, not below 0.9 x the algorithmic bytes)
                try:
                    pmc2 = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["workloads"]
                    e2 = pmc2.get(f"{wname}_x{wrep}", {}).get("kernels", {}).get(gather_kernel())
                except (OSError, ValueError, KeyError):
                    e2 = None
                alg2 = gk["bytes"] / gk["calls"]
                if e2 and world == 1 and e2.get("launches_per_step") and \
                        abs(e2["launches_per_step"] - gk["calls"] / 3) < 1e-6:
                    tr2 = e2["hbm_bytes_per_step"] / e2["launches_per_step"]
                    if tr2 >= 0.9 * alg2:
                        entry["gather"].update({"algorithmic_bytes_per_launch": alg2, "traffic": tr2,
                                                "traffic_over_algorithmic": tr2 / alg2})
            # Scaling evidence one GPU can give: the 8 cost-balanced shards a strong-scaling run over
            # 8 GPUs would hand out (distributed.contiguous_shards on graph_costs: exact neighborhood rows), run one after another HERE; the
            # slowest shard bounds the 8-GPU pass, so (mean shard time) / (slowest shard time) is the efficiency the cost
            # model delivers (inference has no data-path collective: only the final gather of [G, 29] counts is added).
            if world == 1:
                del p2
                torch.cuda.empty_cache()
                cuts = D.contiguous_shards(D.graph_costs(g2, 29, device), 8)
                shard_ms = []
                for lo_, hi_ in cuts:
                    ps = InferencePipeline(nm2, gm2, g2.subset(lo_, hi_), depth=4, device=device,
                                           max_neigh_rows=args.neigh_rows, max_gossip_rows=args.gossip_rows, rank=0, world=1)
                    ps.run()
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        ps.run()
                    sync()
                    shard_ms.append(1e3 * (time.perf_counter() - t0) / 3)
                    del ps
                    torch.cuda.empty_cache()
                entry["strong_scaling_8"] = {
                    "per_shard_ms": [round(v, 3) for v in shard_ms], "graphs_per_shard": [hi_ - lo_ for lo_, hi_ in cuts],
                    "predicted_8gpu_efficiency": (sum(shard_ms) / 8) / max(shard_ms),
                    "single_gpu_ms": entry["ms_per_step"],
                    "note": "8 cost-balanced shards timed one after another on this GPU; efficiency = mean / slowest"}
            else:
                del p2
            secondary[f"{wname}_x{wrep}"] = entry
            del g2
            torch.cuda.empty_cache()
        del nm2, gm2

    graphs_per_step = graphs.num_graphs if strong else graphs.num_graphs * world
    value = graphs_per_step * args.steps / elapsed
    cnt = torch.tensor([pipe.graphs.num_graphs, pipe.graphs.num_nodes, part.num_neigh, part.num_rows,
                        part.num_edges], dtype=torch.int64)
    result = {
        "metric": "graphs/sec (29-query neighborhood+gossip inference)",
        "value": value, "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "arithmetic": "fp32 in/out, fp32 accumulation; matrix products of the SHMP layers, the gossip stage and the anchor "
                      "GEMM, the post_mp tail and the count head's target half as 3 fp16 MFMA products per multiply-add "
                      "(hi/lo split with power-of-two scales), of the table products and post_mp.0 as 6 bf16 products "
                      "(3-way truncation split): both fp32-accurate, DESIGN.md section 4",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}-shaped synthetic ({base.num_graphs} graphs) x{args.replicas} "
                        f"replicas {'in total' if strong else 'per GPU'}, 29 standard queries, depth-4 "
                        f"canonical neighborhoods, neighborhood+gossip inference, random-init weights",
            "graphs_per_gpu": int(cnt[0]), "nodes_per_gpu": int(cnt[1]),
            "neighborhoods_per_gpu": int(cnt[2]), "neighborhood_rows_per_gpu": int(cnt[3]),
            "neighborhood_directed_edges_per_gpu": int(cnt[4]),
            "parallelism": f"dp{world} (graph sharding, no data-path collective)",
            "partition_build_s": round(t_build, 3), "partition_backend": pipe.partition_backend,
            "launch_mode": "hipGraph replay" if args.graph else "eager launches",
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel, from the HIP events of the timed region -----------
        if not args.no_profile:
            summ = primary_summary
            tot = sum(d["ms"] for d in summ.values())
            name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
            calls = d["calls"]
            pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            pmc_all = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
            # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS
            # workload (tools/profile_round.sh; gfx950-corrected), keyed "<workload>_x<replicas>"
            pmc = pmc_all.get("workloads", {}).get(f"{args.workload}_x{args.replicas}", {})

            stale = []

            def traffic(kernel, algorithmic=None, hbm_bound=False):
                """HBM bytes per launch of ``kernel`` from the committed PMC passes -- only if they were
                taken on a tree with the SAME launches per step as this run (else the figure belongs to other
                launches: null + a note) and, for an HBM-bound kernel, not below 0.9 x its algorithmic bytes
                (a kernel cannot move less than it must: such a figure is an artefact)."""
                e = pmc.get("kernels", {}).get(kernel)
                if not e or world != 1 or kernel not in summ:
                    return None
                live = summ[kernel]["calls"] / args.steps
                lps = e.get("launches_per_step")
                if lps is None or abs(lps - live) > 1e-6:
                    stale.append(f"{kernel}: PMC pass has {lps} launches/step, this run {live:g}")
                    return None
                t = e["hbm_bytes_per_step"] / lps
                if hbm_bound and algorithmic and t < 0.9 * algorithmic:
                    stale.append(f"{kernel}: PMC traffic {t:.3e} B/launch below 0.9 x algorithmic {algorithmic:.3e}")
                    return None
                return t

            mp = mfma_peak(name)
            # a kernel that both streams rows and multiplies them (the fused SHMP layer) is priced against the roof it
            # is NEARER to: the binding one.  (The gather launches run at 0.4 of the HBM peak and 0.2 of the f16x3 matrix
            # rate: HBM-bound, as BASELINE.json's north_star names them; the gossip kernel at 0.04 / 0.38: matrix-bound.)
            other_roof = None
            if mp is not None and d["bytes"] > 0:
                f_h = d["bytes"] / (d["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS
                f_m = d["flops"] / (d["ms"] * 1e-3) / 1e12 / mp[0]
                if f_h > f_m:
                    other_roof = {"bound": "mfma", "achieved": f_m * mp[0], "peak": mp[0], "unit": "TFLOP/s", "frac": f_m,
                                  "pipe": mp[1]}
                    mp = None
                else:
                    other_roof = {"bound": "hbm", "achieved": f_h * PEAK_HBM_GBS, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": f_h}
            if mp is not None:
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                roof = {"bound": "mfma", "achieved": ach, "peak": mp[0],
                        "unit": "TFLOP/s", "frac": ach / mp[0], "traffic": traffic(name),
                        "pipe": mp[1],
                        "frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                        "note": "achieved/peak in algorithmic fp32 flops; executed fp16 flops are 3x (frac = "
                                "matrix-pipe utilisation); frac_of_f32_mfma_peak = against what the fp32 matrix "
                                "pipe (157.3 TF/s) could do at best" if mp[0] == PEAK_X3_TFLOPS else
                                "achieved/peak in algorithmic fp32 flops; executed bf16 flops are 6x "
                                "(frac = matrix-pipe utilisation)" if mp[0] == PEAK_X6_TFLOPS else
                                "fp32 matrix pipe"}
            else:
                ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                roof = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": ach / PEAK_HBM_GBS, "traffic": traffic(name, d["bytes"] / calls, hbm_bound=True)}
            if other_roof:
                roof["other_roof"] = other_roof
            roof.update({"kernel": name, "launches": calls, "avg_launch_ms": d["ms"] / calls,
                         "share_of_kernel_time": d["ms"] / tot,
                         "algorithmic_per_launch": (d["flops"] if roof["bound"] == "mfma" else d["bytes"]) / calls,
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / "
                                           "WRITE_SIZE passes, gfx950-corrected)" if roof["traffic"] else None})
            # the north_star's gather = the fused SHMP layer's count-row launches, priced against HBM
            gk = summ.get(gather_kernel())
            if gk and gk["ms"] > 0:
                gbs = gk["bytes"] / (gk["ms"] * 1e-3) / 1e9
                tfs = gk["flops"] / (gk["ms"] * 1e-3) / 1e12
                alg = gk["bytes"] / gk["calls"]
                tr = traffic(gather_kernel(), alg, hbm_bound=True)
                roof["gather"] = {
                    "kernel": gather_kernel(), "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "launches": gk["calls"],
                    "avg_launch_ms": gk["ms"] / gk["calls"], "share_of_kernel_time": gk["ms"] / tot,
                    "algorithmic_bytes_per_launch": alg, "traffic": tr,
                    "traffic_over_algorithmic": None if tr is None else tr / alg,
                    "mfma_TFLOPs": tfs, "mfma_frac_of_split_peak": tfs / gather_mfma_peak(),
                    "mfma_split_peak_TFLOPs": gather_mfma_peak(),
                    "note": "x rows once + out rows once + indices per launch (DESIGN.md section 4); "
                            "the same launches also run the layer's folded GEMM on the matrix pipe"}
            if roof["bound"] == "mfma" and not args.no_attainable:
                att = attainable_mfma()
                if att:
                    # the shape the dominant kernel issues (gossip_fused: 16x16x32; the others: 32x32x16)
                    shape = "v_mfma_f32_16x16x32_bf16" if name in ("gossip_fused_kernel", "gossip_fused_f16_kernel") \
                        else "v_mfma_f32_32x32x16_bf16"      # (the fp16 MFMAs run at the bf16 rate of their shape)
                    if shape in att and mp[0] in (PEAK_X6_TFLOPS, PEAK_X3_TFLOPS):
                        pa = att[shape]["TFLOPs"] / (6.0 if mp[0] == PEAK_X6_TFLOPS else 3.0)
                        roof["attainable"] = {
                            "peak": pa, "frac": roof["achieved"] / pa, "mfma_shape": shape,
                            "measured": att,
                            "note": "bare MFMA loop of that shape on random data on this device, measured in this "
                                    "run (tools/micro/mfma_peak): the chip lowers its clock under dense matrix "
                                    "work, so the datasheet peak (2.4 GHz) is not reachable by any kernel"}
            if stale:
                roof["stale_pmc"] = stale
            result["roofline"] = roof
            result["kernels"] = {
                k: {"calls": v["calls"], "ms": round(v["ms"], 3),
                    "TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 3) if v["ms"] > 0 else None,
                    "GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
            result["kernels_note"] = ("HIP events around every launch of ONE pass in front of the timed region, scaled to its "
                                      "steps; the entry of the dominant kernel (`roofline.kernel`) is from events around its "
                                      "launches INSIDE the timed region")
        # ---- the other BASELINE workload shapes, short runs of the same pass (Syn_1827-shaped: C3 / C4,
        #      MSRC-21 + IMDB-BINARY-shaped: C5), each with its own gather roofline --------------------
        if not args.no_secondary:
            result["secondary"] = secondary
        # ---- small-dataset latency: the real dataset size (x1), hipGraph replay vs eager --------
        if world == 1 and not args.no_x1 and args.replicas != 1:
            p1 = InferencePipeline(nm, gm, base, depth=4, device=device)
            lat = {}
            for mode, fn in (("eager", p1.run), ("hipgraph_replay", p1.step)):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(50):
                    fn()
                torch.cuda.synchronize(device)
                dt = (time.perf_counter() - t0) / 50
                lat[mode] = {"ms_per_pass": 1e3 * dt, "graphs_per_s": base.num_graphs / dt}
            lat["graphs"] = base.num_graphs
            lat["note"] = ("one pass over the un-replicated dataset; InferencePipeline.step() replays a "
                           "hipGraph for shards under 400k neighborhood rows (launch-bound otherwise)")
            result["latency_x1"] = lat
        # ---- training leg (BASELINE configs 3 / 4), N=1 only -------------------------------------------------
        if world == 1 and not args.no_train:
            del pipe
            torch.cuda.empty_cache()
            # fp32 (the parity-tested default) AND bf16 (BASELINE config 3 names bf16) -- each with its own roofline --,
            # then the gossip stage of config 4
            precisions = ["fp32", "bf16"] if args.train_precision == "both" else [args.train_precision]
            legs = {p_: train_leg(device, stride=args.train_stride, precision=p_) for p_ in precisions}
            result["train_syn_1827"] = legs[precisions[0]]
            for p_ in precisions[1:]:
                result["train_syn_1827"][p_] = legs[p_]
            torch.cuda.empty_cache()
            # the same steps through the data-parallel path against real RCCL, world of one rank (a fresh child process)
            ddp = ddp_world1_leg(stride=args.train_stride)
            if "ms_per_step" in ddp:
                ddp["vs_single_process_replay"] = ddp["ms_per_step"] / result["train_syn_1827"]["ms_per_step"]
            result["train_syn_1827"]["ddp_world1_nccl"] = ddp
            result["train_gossip"] = train_gossip_leg(device)
        # ---- CPU baseline (N=1 only) + parity of the sample ------------------------------------
        if world == 1 and not args.no_cpu_baseline:
            with open(os.path.join(ROOT, "tests", "golden", "queries.json")) as f:
                qj = json.load(f)
            queries = [(q["n"], [tuple(e) for e in q["edges"]]) for q in qj["queries"]]
            cb, ref, n = cpu_baseline(nm, gm, base.edge_lists(), queries, args.cpu_seconds, workers=cpu_workers)
            m = min(n, base.num_graphs)
            got = out["graph_gossip_count"][:m].cpu()
            want = ref["graph_gossip_count"][:m]
            cb["max_abs_diff_vs_gpu"] = (got - want).abs().max().item()
            # graph-level counts reach 1e8 on the dense shapes (sums of 2**logit - 1): the parity figure is relative
            cb["max_rel_diff_vs_gpu"] = ((got - want).abs() / (1.0 + want.abs())).max().item()
            cb["max_abs_count"] = want.abs().max().item()
            result["cpu_baseline"] = cb
            result["gpu_over_cpu"] = value / cb["value"]
        if collective is not None:
            result["collective"] = collective
        print(json.dumps(result))
    stop_cpu_workers(cpu_workers)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
