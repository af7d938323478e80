#!/usr/bin/env python3
"""Headline benchmark: RandLA-Net training clouds/sec on MI355X (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json `metric`: "training clouds/sec, N=40960 pts, bs=8"): N=40960 points per
cloud, 2 classes, k=16, encoder layers [16,64,128,256], 8 clouds per GPU, synthetic clouds
xyz ~ U[0,1)^3 from RandomState(1234+rank), labels = sphere rule, random-init weights.  One step = per-forward
numpy permutation -> forward -> dice loss + metric counts -> backward -> (RCCL all-reduce of
the flat gradient when N>1) -> Adam, all on hand-written HIP kernels (librandla_hip.so),
replayed as a hipGraph.  Inputs are resident in HBM before the timed region.  Weak scaling:
per-GPU batch is fixed, clouds are sharded over ranks, the only collective is one all-reduce.

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (a parent that makes
no GPU call runs `python -m torch.distributed.run`); fewer than N visible devices is an error, never a
silent one-GPU run.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  config_A     - the same step at BASELINE.json config A / B (4 clouds per GPU, weak scaling)
  strong_bs8   - (N > 1) the metric's global batch of 8 split over the N ranks
  bf16_operands- the metric's configuration with plain bf16 operands in the wide kernels (BASELINE config A says "bf16");
                 storage stays fp32 and its logits leave the 1e-3 parity bound, so it is never `value`
  fp32_exact   - the metric's configuration with exact fp32 products everywhere (RL_WIDE_GEMM=fp32: the reference's own
                 arithmetic)
  bf16_storage - the metric's configuration with the neighbourhood-row gradient tensors stored as bf16 (rl_set_storage)
  config_S / config_Kt_shard - BASELINE configs[3] (65536 pts, 13 classes, 5 layers, bs=8) and the per-GPU shard of
                 configs[4] (122880 pts, 20 classes, bs=2), each with clouds/s and its whole-step roofline fraction
  whole_step / knn / mfma_by_level - path-level roofline figures (also kept inside `roofline`)
  roofline     - the kernel with the largest share of the step (measured with HIP events around
                 every launch of an instrumented eager pass on the launch stream): algorithmic
                 bytes per launch / mean launch duration vs the 8 TB/s HBM peak.
  cpu_baseline - the same training step on the host cores: oracle/ restatement of the
                 reference's PyTorch-CPU graph + single-threaded exact C KNN ("port"), on a
                 bounded sample (B=2 clouds, 1 warm-up + 3 timed steps).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "3d_recognizer_amd"))
sys.path.insert(0, REPO)

import torch  # noqa: E402

CFG = dict(n_points=40960, n_classes=2, n_neighbors=16, layer_sizes=[16, 64, 128, 256], per_gpu_batch=8)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2516.6  # same guide: ~2.5 PF dense = 16x the fp32 matrix rate
# arithmetic of the wide (K or N > 64) GEMM / weight-gradient kernels, chosen by librandla_hip from this variable:
# "bf16x3" (default): fp32 operands split into bf16 head + tail, three bf16 MFMAs per product, fp32 accumulate
# (logits within 2.5e-6 of the oracle on configs A/S/Kt, tests/test_configs_gpu.py); "fp32": v_mfma_f32_16x16x4_f32;
# "bf16": heads only (throughput mode, 1e-3 logits parity NOT met)
WIDE_GEMM = os.environ.get("RL_WIDE_GEMM", "bf16x3")   # main() checks it against what the library reports
WIDE_KERNELS = ("wgemm_kernel", "wgemm2_kernel", "pgemm_kernel", "pwgrad128")


def synthetic_batch(B, N, C, seed):
    """xyz ~ U[0,1)^3; class = 1 + floor((C-1) z) clipped inside a sphere of radius
    0.25 around the centre, else 0."""
    rs = np.random.RandomState(seed)
    xyz = rs.uniform(0.0, 1.0, (B, N, 3)).astype(np.float32)
    inside = np.linalg.norm(xyz - 0.5, axis=-1) < 0.25
    cls = np.clip(1 + np.floor((C - 1) * xyz[..., 2]).astype(np.int64), 1, C - 1)
    return xyz, np.where(inside, cls, 0).astype(np.int64)


# the other single-GPU BASELINE.json configurations, reported beside the metric's (never as `value`)
CFG_S = dict(n_points=65536, n_classes=13, n_neighbors=16, layer_sizes=[16, 64, 128, 256, 512], per_gpu_batch=8)
CFG_KT = dict(n_points=122880, n_classes=20, n_neighbors=16, layer_sizes=[16, 64, 128, 256], per_gpu_batch=2)


def build_model(device, seed=0, cfg=None):
    from randlanet.utils.modules import RandLANet, RandLANetSettings
    cfg = CFG if cfg is None else cfg
    torch.manual_seed(seed)
    s = RandLANetSettings(n_classes=cfg["n_classes"], n_points=cfg["n_points"], n_neighbors=cfg["n_neighbors"],
                          layer_sizes=list(cfg["layer_sizes"]), knn="kdtree")
    return RandLANet(s, device)


def host_cores():
    """Cores this process may use: the scheduler affinity, capped at the 16-core CPU share a
    one-GPU box grants (oversubscribing the 256 visible cores made torch-CPU ~30x slower)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("RL_CPU_BASELINE_CORES", "16"))))


def cpu_baseline(steps=3, B=2):
    """The reference's training step on the host: PyTorch-CPU NCHW graph + exact C KNN (oracle/)."""
    from oracle import randlanet_oracle as O
    from oracle.loss_metrics_oracle import loss_by_name
    cores = host_cores()
    torch.set_num_threads(cores)
    net = build_model(torch.device("cpu"))
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    params = [v.requires_grad_(True) for k, v in P.items() if v.is_floating_point() and "running" not in k]
    opt = torch.optim.Adam(params, lr=1e-2)
    xyz, labels = synthetic_batch(B, CFG["n_points"], CFG["n_classes"], 1234)
    x, y = torch.from_numpy(xyz), torch.from_numpy(labels)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        perm = np.random.permutation(CFG["n_points"])
        buffers = {}
        logits = O.forward(P, x, perm, layer_sizes=CFG["layer_sizes"], n_neighbors=CFG["n_neighbors"],
                           training=True, buffers=buffers)
        loss = loss_by_name("dice", logits, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for k, v in buffers.items():
            P[k] = v
        times.append(time.perf_counter() - t0)
    t = float(np.mean(times[1:]))
    return dict(value=round(B / t, 4), unit="clouds/s", cores=cores, kind="port",
                sample=f"{steps} timed steps (+1 warm-up) of the same train step at B={B} clouds of "
                       f"{CFG['n_points']} points; PyTorch-CPU NCHW restatement (oracle/randlanet_oracle.py) "
                       "+ single-threaded exact C KNN (oracle/knn_oracle.c)")


MIN_WARMUP_S = 0.4      # untimed steps run for at least this long before a timed window (clock ramp)
METRIC = "training clouds/sec, N=40960 pts, bs=8, at 1/2/4/8 GPUs; mIoU parity"     # BASELINE.json `metric`, verbatim


def fused_min_bytes_per_cloud(N, K, layers, C, e, dec=4):
    """SURVEY.md 8(d) "network fused-minimum HBM bytes" of one FORWARD of one cloud with e-byte activations (A: 103 MB
    at e = 2, 190 MB at e = 4); a training step is priced at 3x that (forward + reload in backward + gradient traffic)."""
    tot, n_in, n = 0, 8, N
    for d in layers:
        tot += 12 * n + 4 * n * K + e * n_in * n + 12 * n * K + 2 * e * (d // 2) * n * K + 2 * e * (d // 2) * n + e * 2 * d * n
        n_in, n = 2 * d, n // dec
    L = len(layers)
    cin = 4 * layers[-1]
    for j in range(L):
        n *= dec
        cout = 8 if j == L - 1 else 2 * layers[L - 2 - j]
        tot += e * (cin + cout) * n + 4 * n
        cin = 2 * cout
    return tot + 4 * C * N


MFMA_KERNELS = ("wgemm_kernel", "wgemm2_kernel", "pgemm_kernel", "pwgrad", "wgrad_kernel", "sgemm_kernel", "swgrad_kernel", "gemm_kernel",
                "pool_fwd_kernel", "pool_bwd_kernel", "pool128_bwd_kernel", "rpe_wgrad_kernel", "rpe_stats_kernel",
                "vpool_fwd_kernel", "vpool_bwd_kernel", "vrpe_wgrad_kernel", "vrpe_stats_kernel")
# the fused tile kernels: bf16x3 products on the bf16 MFMAs like the wide kernels, but bound by neither roof - by instruction
# issue and the latency of their gathers; the line says so next to the two fractions
TILE_KERNELS = ("vpool_fwd_kernel", "vpool_bwd_kernel", "vrpe_wgrad_kernel", "vrpe_stats_kernel", "vrpe_bn_reduce_kernel",
                "pool_fwd_kernel", "pool_bwd_kernel", "pool128_bwd_kernel", "attpool_fwd16_kernel", "attpool_bwd16_kernel")


def function_name(kernel: str) -> str:
    """Kernel FUNCTION of a launch: the name without its template arguments - every instantiation of a function is summed,
    the way `rocprofv3 --stats` rows are summed per function when the dominant one is picked.  A split-K launch is reported
    by the library as "<function>+splitk" (its event window also holds the few-microsecond reducer): it counts under the function."""
    return kernel.split("<")[0].split("+")[0].strip()


def function_roofline(name, f):
    """bound / achieved / peak / frac of one kernel function from its algorithmic bytes and flops per step and its time."""
    secs = max(f["ms"], 1e-9) * 1e-3
    ai = f["flops"] / max(f["bytes"], 1)
    if (name.startswith(WIDE_KERNELS) or name.startswith(TILE_KERNELS)) and WIDE_GEMM != "fp32":
        mfma_peak, spent = BF16_MFMA_PEAK_TFLOPS, (1 if WIDE_GEMM == "bf16" else 3)
    else:
        mfma_peak, spent = F32_MFMA_PEAK_TFLOPS, 1
    if name.startswith(MFMA_KERNELS) and spent * ai > mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9):
        achieved, peak, unit, bound = f["flops"] / secs / 1e12, mfma_peak, "TFLOP/s", "mfma"
    else:
        achieved, peak, unit, bound = f["bytes"] / secs / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
    out = dict(bound=bound, achieved=round(achieved, 2), peak=peak, unit=unit, frac=round(achieved / peak, 5),
               hbm_frac=round(f["bytes"] / secs / 1e9 / HBM_PEAK_GBS, 5),
               mfma_frac=round(spent * f["flops"] / secs / 1e12 / mfma_peak, 5))
    if name.startswith(TILE_KERNELS):
        out["limited_by"] = "instruction issue / gather latency (neither roof): see DESIGN.md section 5"
    return out


def roofline_pass(stepper, eager_steps=3):
    """Instrumented eager steps: HIP events around every launch, on the launch stream.

    Launches are grouped the way `rocprofv3 --kernel-trace --stats` groups them - by kernel FUNCTION
    (librandla_hip reports which one each entry point dispatched to) - so the numbers can be held against the
    committed rocprof summary.  The DOMINANT kernel is the function with the largest share of the step.  Its
    `achieved` = algorithmic bytes (or flops) of its launches / their measured time, i.e. per-launch average over
    per-launch average; its bound follows its arithmetic intensity against the ridge point of the matrix unit it
    runs on (fp32 MFMA 157.3 TFLOP/s / 8 TB/s = 19.7 flop/B; the wide kernels in bf16x3 mode spend 3 bf16 MFMA flops
    per algorithmic flop against 2516.6 TFLOP/s: ridge 105 algorithmic flop/B, so K = N = 128 layers are HBM-bound).  Per shape the median over the eager steps is used."""
    from randlanet import _ops as ops
    ops.TIMER = ops.KernelTimer()
    g_main, g_adam = stepper._g_main, stepper._g_adam
    stepper._g_main = stepper._g_adam = None
    from randlanet import _hip as _H
    try:
        n0 = _H.lib().rl_launch_count()
        for _ in range(eager_steps):
            stepper.step(np.random.permutation(stepper.N))
        torch.cuda.synchronize()
        kernel_launches = (_H.lib().rl_launch_count() - n0) / eager_steps
        records = ops.TIMER.records
    finally:
        ops.TIMER = None
        stepper._g_main, stepper._g_adam = g_main, g_adam
    shapes = {}
    for cat, key, nbytes, flops, e0, e1, kern, lvl in records:
        a = shapes.setdefault((kern, cat, key, lvl), dict(kernel=kern, category=cat, shape=key, times=[], nb=[], fl=[], level=lvl))
        a["times"].append(e0.elapsed_time(e1))
        a["nb"].append(nbytes)
        a["fl"].append(flops)
    rows = []
    for a in shapes.values():
        # (launches that share an op string - the two pooling stages of a level - differ in their algorithmic bytes: the mean)
        a["bytes"], a["flops"] = float(np.mean(a["nb"])), float(np.mean(a["fl"]))
        per_launch = float(np.median(a["times"]))
        launches = len(a["times"]) / eager_steps
        rows.append(dict(kernel=a["kernel"], category=a["category"], shape=a["shape"], level=a["level"], ms_per_launch=per_launch,
                         launches_per_step=launches, ms_per_step=per_launch * launches, bytes=a["bytes"], flops=a["flops"]))
    rows.sort(key=lambda r: -r["ms_per_step"])
    total_ms = sum(r["ms_per_step"] for r in rows)
    funcs = {}
    for r in rows:
        f = funcs.setdefault(function_name(r["kernel"]), dict(ms=0.0, launches=0.0, bytes=0.0, flops=0.0))
        f["ms"] += r["ms_per_step"]; f["launches"] += r["launches_per_step"]
        f["bytes"] += r["bytes"] * r["launches_per_step"]; f["flops"] += r["flops"] * r["launches_per_step"]
    name, top = max(funcs.items(), key=lambda kv: kv[1]["ms"])
    fr = function_roofline(name, top)
    big = max((r for r in rows if function_name(r["kernel"]) == name), key=lambda r: r["ms_per_step"])
    roof = dict(bound=fr["bound"], kernel=name, achieved=fr["achieved"], peak=fr["peak"], unit=fr["unit"],
                frac=fr["frac"], traffic=None,
                avg_launch_us=round(1e3 * top["ms"] / top["launches"], 2), launches_per_step=round(top["launches"], 1),
                bytes_per_launch=int(top["bytes"] / top["launches"]), flops_per_launch=int(top["flops"] / top["launches"]),
                share_of_step=round(top["ms"] / total_ms, 3),
                largest_shape=dict(op=f"{big['category']}{list(big['shape'])}", us_per_launch=round(1e3 * big["ms_per_launch"], 1),
                                   TFLOPs=round(big["flops"] / max(big["ms_per_launch"], 1e-9) / 1e9, 2),
                                   GBps=round(big["bytes"] / max(big["ms_per_launch"], 1e-9) / 1e6, 1)))
    for k in ("hbm_frac", "mfma_frac", "limited_by"):
        if k in fr:
            roof[k] = fr[k]
    # the five largest kernel functions of the step, each against its own roof
    roof["by_function"] = []
    for fname, f in sorted(funcs.items(), key=lambda kv: -kv[1]["ms"])[:5]:
        e = dict(kernel=fname, ms_per_step=round(f["ms"], 4), share=round(f["ms"] / total_ms, 4),
                 launches_per_step=round(f["launches"], 1))
        e.update(function_roofline(fname, f))
        roof["by_function"].append(e)
    # PMC traffic (profiles/pmc_traffic.json, rocprofv3 --pmc passes of the committed profile): keyed by "<function>|<op>" per
    # LAUNCH SHAPE, so that the bytes printed here belong to the shape printed beside them; the ratio to that launch's
    # algorithmic bytes, and the same ratio over all launches of the function ("<function>|*": sum PMC / sum algorithmic)
    pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            table = json.load(open(pmc))
            key = f"{name}|{roof['largest_shape']['op']}"
            if key in table:
                roof["traffic"] = table[key]
                roof["traffic_shape"] = roof["largest_shape"]["op"]
                # (the PMC pass measured ONE launch form of the shape - for a GEMM the plain one, without the accumulating read of Y
                # that some of the shape's launches in a step carry: its algorithmic bytes ride along in the table where they differ)
                roof["traffic_over_algorithmic"] = round(table[key] / max(table.get(key + "|algorithmic", big["bytes"]), 1), 3)
            tot = table.get(f"{name}|*")
            if tot is not None:
                roof["traffic_function_per_step"] = tot
                roof["traffic_function_over_algorithmic"] = round(tot / max(top["bytes"], 1), 3)
        except Exception:
            pass
    roof["kernel_launches_per_step"] = round(kernel_launches, 1)
    breakdown = dict(step_kernel_ms=round(total_ms, 3), kernel_launches_per_step=round(kernel_launches, 1),
                     kernels={k: dict(ms_per_step=round(v["ms"], 3), launches_per_step=round(v["launches"], 1),
                                      avg_launch_us=round(1e3 * v["ms"] / max(v["launches"], 1e-9), 1),
                                      GBps=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                                      TFLOPs=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2))
                              for k, v in sorted(funcs.items(), key=lambda kv: -kv[1]["ms"])},
                     top_shapes=[dict(kernel=r["kernel"], op=f"{r['category']}{list(r['shape'])}",
                                      ms_per_step=round(r["ms_per_step"], 3), launches_per_step=r["launches_per_step"],
                                      GBps=round(r["bytes"] / max(r["ms_per_launch"], 1e-9) / 1e6, 1),
                                      TFLOPs=round(r["flops"] / max(r["ms_per_launch"], 1e-9) / 1e9, 2)) for r in rows[:80]],
                     # consumed by step_rooflines(), not printed
                     all_shapes=[dict(op=f"{r['category']}{list(r['shape'])}", level=r["level"], ms_per_step=r["ms_per_step"],
                                      launches_per_step=r["launches_per_step"], bytes=r["bytes"], flops=r["flops"]) for r in rows])
    return roof, breakdown


def visible_gpu_count(sysfs="/sys/class/kfd/kfd/topology/nodes") -> int:
    """GPUs this process could use, WITHOUT any HIP call: KFD topology nodes with SIMDs (CPU nodes have simd_count 0),
    narrowed by the *_VISIBLE_DEVICES lists when set."""
    n = 0
    try:
        for node in sorted(os.listdir(sysfs)):
            try:
                with open(os.path.join(sysfs, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as fresh children.  This parent makes no HIP call
    at all (the device count comes from sysfs), so nothing that has initialised the GPU is ever re-executed."""
    import socket
    import subprocess
    have = visible_gpu_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} device(s) are visible - refusing to measure fewer GPUs "
              "than asked for", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_steps(stepper, N, steps, warmup, barrier, world, dist, dev):
    """W untimed steps, then exactly K steps between barrier + synchronize; MAX over ranks.  Returns seconds."""
    tw = time.perf_counter()
    for _ in range(warmup):
        stepper.step(np.random.permutation(N))
    # ... and at least MIN_WARMUP_S seconds of it whatever --warmup says: a 20-step window right after a cold start was seen
    # 8 % low (clocks still ramping)
    extra = 0
    while True:
        barrier()
        el = time.perf_counter() - tw
        if world > 1:      # one decision for all ranks (every step contains a collective)
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            el = float(t.item())
        if el >= MIN_WARMUP_S or extra >= 2000:
            break
        for _ in range(10):
            stepper.step(np.random.permutation(N))
        extra += 10
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        stepper.step(np.random.permutation(N))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    m = stepper.last_metrics()
    if not np.isfinite(m["loss"]):
        raise SystemExit("bench.py: the training loss is not finite - the measurement is invalid")
    return elapsed, m


def whole_step_roofline(cfg, per_gpu_clouds_per_s, e=4):
    """The whole step against SURVEY.md 8(d)'s fused-minimum HBM traffic (3x the forward's, e-byte activations)."""
    min_bytes = 3 * fused_min_bytes_per_cloud(cfg["n_points"], cfg["n_neighbors"], cfg["layer_sizes"], cfg["n_classes"], e)
    return {"bound": "hbm", "fused_min_bytes_per_cloud": min_bytes, "storage": "f32" if e == 4 else "bf16",
            "achieved": round(per_gpu_clouds_per_s * min_bytes / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(per_gpu_clouds_per_s * min_bytes / 1e9 / HBM_PEAK_GBS, 4)}


def step_rooflines(breakdown, B, value, world):
    """Path-level roofline figures from the instrumented pass: the whole step against the fused-minimum
    HBM traffic, the neighbour search two ways, and the matrix work per encoder level."""
    layers = CFG["layer_sizes"]
    ws = whole_step_roofline(CFG, value / world)
    # every kernel the library launches in a step (its own counter: what a rocprofv3 kernel trace counts for the same step,
    # without the runtime's few copy kernels) - not just the launches the event timer brackets
    ws["launches_per_step"] = breakdown["kernel_launches_per_step"]
    out = {"whole_step": ws}
    rows = breakdown["all_shapes"]
    knn = [r for r in rows if r["op"].startswith("knn")]
    if knn:
        ms = sum(r["ms_per_step"] for r in knn)
        nbytes = sum(r["bytes"] * r["launches_per_step"] for r in knn)
        pairs = sum(r["flops"] * r["launches_per_step"] for r in knn) / 8.0
        out["knn"] = {"ms_per_step": round(ms, 3), "io_GBps": round(nbytes / ms / 1e6, 1),
                      "io_frac_of_hbm": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                      "brute_equivalent_pair_evals_per_s": round(pairs / ms * 1e3, 0),
                      "note": "exact grid search: pairs are the brute-force count the answer is equivalent to"}
    # matrix work per encoder level (the engine tags every launch with the level it works on)
    lv = []
    for l, d in enumerate(layers):
        sel = [r for r in rows if r["flops"] > 0 and r["level"] == l]
        fl = sum(r["flops"] * r["launches_per_step"] for r in sel)
        ms = sum(r["ms_per_step"] for r in sel)
        if ms > 0:
            lv.append({"level": l, "d": d, "GFLOP_per_step": round(fl / 1e9, 2), "ms_per_step": round(ms, 3),
                       "TFLOPs": round(fl / ms / 1e9, 1), "frac_fp32_mfma_peak": round(fl / ms / 1e9 / F32_MFMA_PEAK_TFLOPS, 4),
                       "frac_bf16_mfma_peak_x3": round(3 * fl / ms / 1e9 / BF16_MFMA_PEAK_TFLOPS, 4),
                       "level_ms_all_kernels": round(sum(r["ms_per_step"] for r in rows if r["level"] == l), 3)})
    out["mfma_by_level"] = lv
    return out


def trainer_e2e(dev, clouds=400, batch=4):
    """Clouds/s through the callers' side of the path: Model.train -> Trainer.train -> device data loader
    (one rl_batch_assemble launch per batch) -> TrainStep graph replay, config A shape (4 clouds of 40960 points per step),
    default augmentation, one timed epoch of clouds/batch steps after a warm-up epoch (graph capture, allocator).  The epoch
    is timed between the Trainer's per-epoch callbacks, so it contains that epoch's validation (one batch x 10 passes).
    Two random-number modes of the loader: "numpy" (the reference's streams, drawn on the host: ~3.5 ms of numpy per cloud
    bounds it) and "device" (sample indices and jitter noise from a device generator)."""
    import logging
    from randlanet import Model, RandLANetSettings, TrainingSettings
    logging.getLogger("trainer").setLevel(logging.WARNING)
    N, C = CFG["n_points"], CFG["n_classes"]
    rs = np.random.RandomState(7)
    n_raw = N + 4096

    def cloud():
        xyz = rs.uniform(0.0, 1.0, (n_raw, 3)).astype(np.float32)
        lab = (np.linalg.norm(xyz - 0.5, axis=-1) < 0.25).astype(np.int64)
        return xyz, np.zeros((n_raw, 0), np.float32), lab
    train = [cloud() for _ in range(clouds)]
    val = train[:batch]
    out = {"workload": f"Model.train, {clouds} clouds of {n_raw} points sampled to {N}, bs={batch}, default augmentation, "
                       f"one epoch = {clouds // batch} steps + validation (1 batch x 10 passes); device data loader", "unit": "clouds/s"}
    for mode in ("numpy", "device"):
        os.environ["RL_PIPELINE_RNG"] = mode
        try:
            torch.manual_seed(0)
            np.random.seed(0)
            model = Model(RandLANetSettings(n_classes=C, n_points=N, n_neighbors=CFG["n_neighbors"],
                                            layer_sizes=list(CFG["layer_sizes"])))
            stamps = []
            settings = TrainingSettings(epochs=2, batch_size=batch, early_stopping=False)
            model.train(train, val, settings, class_names=[f"c{i}" for i in range(C)],
                        callbacks=[lambda e, m: (torch.cuda.synchronize(dev), stamps.append(time.perf_counter()))])
            dt = stamps[1] - stamps[0]
            out[f"rng_{mode}"] = {"value": round(clouds / dt, 2), "ms_per_step": round(1e3 * dt / (clouds // batch), 3)}
            del model
        finally:
            os.environ.pop("RL_PIPELINE_RNG", None)
        torch.cuda.empty_cache()
    return out


def train_py_setting(dev, steps=100, warmup=10):
    """The setting the reference's own CLI trains with (reference train.py:50-59: n_points 2500, n_neighbors 32, batch 4,
    2 classes, 4 layers): K = 32 takes the UN-FUSED path on every level (gather + score GEMM + softmax-pool kernels; the fused /
    virtual tile kernels are written for 16 neighbours).  One replayed step graph, inputs resident in HBM."""
    from randlanet._train import TrainStep
    cfg = dict(n_points=2500, n_classes=2, n_neighbors=32, layer_sizes=[16, 64, 128, 256], per_gpu_batch=4)
    m = build_model(dev, seed=0, cfg=cfg)
    m.train()
    B, N = cfg["per_gpu_batch"], cfg["n_points"]
    st = TrainStep(m, B, N, loss="dice", lr=1e-2)
    x, y = synthetic_batch(B, N, cfg["n_classes"], 99)
    st.set_batch(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    st.capture()
    for _ in range(warmup):
        st.step(np.random.permutation(N))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(np.random.permutation(N))
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    m_ = st.last_metrics()
    if not np.isfinite(m_["loss"]):
        raise SystemExit("bench.py: the train.py-setting step produced a non-finite loss")
    return {"workload": "reference train.py:50-59 setting: 2500 pts/cloud, K=32, bs=4, 2 classes, 4 layers (un-fused pooling path)",
            "value": round(B * steps / dt, 1), "unit": "clouds/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
            "final_loss": round(m_["loss"], 5)}


def predict_latency(reps=20, n_cloud=120000):
    """Config P (BASELINE.json configs[0], reference predict.py:22-31 / the UI's 250 ms timer main.py:49): ms per Model.predict
    on one mock-shaped cloud - seed-0 down-sample to 2500 points, forward (K = 32, 4 layers), full-resolution nearest-neighbour
    up-sampling ("nni"), softmax confidences back on the host."""
    from randlanet import Model, RandLANetSettings
    rs = np.random.RandomState(0)
    cloud = rs.rand(n_cloud, 3).astype(np.float32)
    model = Model(RandLANetSettings(n_classes=2, n_points=2500, n_neighbors=32, upsampling="nni"))
    for _ in range(3):
        conf = model.predict(cloud)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        conf = model.predict(cloud)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    if conf.shape != (2, n_cloud) or not np.isfinite(conf).all():
        raise SystemExit("bench.py: Model.predict returned a malformed result")
    return {"workload": f"Model.predict: one cloud of {n_cloud} points -> 2500 sampled, K=32, 4 layers, nni up-sampling to every "
                        "input point, confidences on the host", "value": round(1e3 * dt, 3), "unit": "ms per cloud", "reps": reps,
            "higher_is_better": False}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~1.7 s timed after ~0.2 s of warm-up - a 30-step (0.25 s) window right after a cold start was seen 8 % low
    # once (clocks still ramping); the whole default run stays well under a minute
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="skip the secondary eval-forward measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config_A / strong_bs8 / bf16_operands / fp32_exact objects")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config_S / config_Kt_shard objects")
    ap.add_argument("--no-callers", action="store_true", help="skip the trainer_e2e / predict_P objects (the callers' side of the path)")
    ap.add_argument("--batch", type=int, default=CFG["per_gpu_batch"], help="clouds per GPU (the metric's bs=8)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank control flow on a one-GPU box)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.set_num_threads(max(1, host_cores() // world))   # a one-GPU box shows 256 cores but grants 16: keep host ops cheap
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} device(s): one GPU per rank is required")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        world = dist.get_world_size()            # n_gpus in the line = what the backend reports

    from randlanet._train import InferStep, TrainStep, broadcast_flat
    from randlanet import _ops as _o
    global WIDE_GEMM
    WIDE_GEMM = _o.get_wide_gemm()             # what the kernels will really do
    B, N, C = args.batch, CFG["n_points"], CFG["n_classes"]
    model = build_model(dev, seed=0)          # identical replicas: same seed on every rank
    model.train()
    stepper = TrainStep(model, B, N, loss="dice", lr=1e-2, use_graph=not args.no_graph, world_size=world)
    xyz, labels = synthetic_batch(B, N, C, 1234 + rank)
    x_dev, y_dev = torch.from_numpy(xyz).to(dev), torch.from_numpy(labels).to(dev)
    stepper.set_batch(x_dev, y_dev)
    np.random.seed(1234 + rank)                # rank-distinct permutation streams
    broadcast_flat(stepper.flat.param, world)  # replicas must start identical
    stepper.capture()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    elapsed, metrics = timed_steps(stepper, N, args.steps, args.warmup, barrier, world, dist, dev)
    value = B * world * args.steps / elapsed

    # secondary configurations: the same model / optimiser state, other batch shapes
    def secondary(per_gpu: int, scaling: str, what: str):
        st = TrainStep(model, per_gpu, N, loss="dice", use_graph=not args.no_graph, state=stepper.state)
        st.set_batch(x_dev[:per_gpu].contiguous(), y_dev[:per_gpu].contiguous())
        st.capture()
        el, m = timed_steps(st, N, args.steps, args.warmup, barrier, world, dist, dev)
        return {"workload": what, "value": round(per_gpu * world * args.steps / el, 3), "unit": "clouds/s",
                "per_gpu_batch": per_gpu, "global_batch": per_gpu * world, "scaling": scaling,
                "ms_per_step": round(1e3 * el / args.steps, 3), "final_loss": round(m["loss"], 5)}
    config_a = strong = sweep = None
    if not args.no_secondary:
        if B != 4:
            config_a = secondary(4, "weak", "BASELINE.json config A (1 GPU) / B (8 GPUs): 4 clouds per GPU")
        if world == 1 and B == CFG["per_gpu_batch"]:
            # the compute side of the metric's "bs=8 at 1/2/4/8 GPUs" read as strong scaling: the step of ONE GPU's shard
            sweep = {"what": "ms per step of one GPU at per-GPU batch 1, 2, 4, 8 (one MI355X; no collective)", "ms_per_step": {}}
            for b in (1, 2):
                sweep["ms_per_step"][str(b)] = secondary(b, "weak", "")["ms_per_step"]
            sweep["ms_per_step"]["4"] = config_a["ms_per_step"] if config_a else None
            sweep["ms_per_step"]["8"] = round(1e3 * elapsed / args.steps, 3)
            a, b8 = sweep["ms_per_step"]["4"], sweep["ms_per_step"]["8"]
            if a:
                sweep["marginal_ms_per_cloud"] = round((b8 - a) / 4.0, 4)
                sweep["fixed_ms_per_step"] = round(b8 - 8 * (b8 - a) / 4.0, 4)
        if world > 1 and 8 % world == 0 and 8 // world != B:
            strong = secondary(8 // world, "strong", "the metric's global batch of 8 clouds split over the ranks")
    # BASELINE config A names "bf16": the same step with plain bf16 operands in the wide GEMMs / weight gradients
    # (rl_set_wide_gemm("bf16"); storage and accumulation stay fp32).  Outside the 1e-3 logit bound (0.7e-3 ... 1.1e-3,
    # DESIGN.md section 5), so it is reported beside the parity mode, never as `value`.
    bf16_ops = None
    if not args.no_secondary and WIDE_GEMM == "bf16x3":
        _o.set_wide_gemm("bf16")
        try:
            bf16_ops = secondary(B, "weak", "the metric's configuration with bf16 (1-term) operands in the wide kernels, fp32 storage")
            bf16_ops["dtype"] = "f32 storage/accumulate, bf16 MFMA operands in the wide GEMMs (not the parity mode)"
        finally:
            _o.set_wide_gemm("bf16x3")
    # ... and with the reference's own arithmetic: exact fp32 products in every kernel
    fp32_exact = None
    if not args.no_secondary and WIDE_GEMM == "bf16x3":
        _o.set_wide_gemm("fp32")
        try:
            fp32_exact = secondary(B, "weak", "the metric's configuration with exact fp32 products in every kernel (RL_WIDE_GEMM=fp32)")
            fp32_exact["dtype"] = "f32"
            fp32_exact["whole_step"] = whole_step_roofline(CFG, fp32_exact["value"] / world)
        finally:
            _o.set_wide_gemm("bf16x3")

    # BASELINE config A's "bf16" as far as it pays here: bf16 STORAGE of the neighbourhood-row gradient tensors (GU / DG of
    # the fused pooling blocks, level-2 X / dS); arithmetic as in the parity mode.  profiles/r03_bf16_storage_experiment.md
    # holds what storing every activation as bf16 measured (no net gain with these kernels) and why.
    bf16_storage = None
    if not args.no_secondary and WIDE_GEMM == "bf16x3" and _o.get_storage() == "f32":
        _o.set_storage("bf16")
        try:
            bf16_storage = secondary(B, "weak", "the metric's configuration with bf16 storage of the neighbourhood-row gradient tensors")
            bf16_storage["dtype"] = "bf16 storage of GU / DG / level-2 X, dS; f32 activations and accumulators, bf16x3 MFMA"
            bf16_storage["whole_step"] = whole_step_roofline(CFG, bf16_storage["value"] / world)
        finally:
            _o.set_storage("f32")

    # the other single-GPU BASELINE.json configurations: their own model, a shorter window (they are not the metric)
    def other_config(cfg, what):
        m = build_model(dev, seed=0, cfg=cfg)
        m.train()
        Bc, Nc = cfg["per_gpu_batch"], cfg["n_points"]
        st = TrainStep(m, Bc, Nc, loss="dice", lr=1e-2, use_graph=not args.no_graph, world_size=world)
        xc, yc = synthetic_batch(Bc, Nc, cfg["n_classes"], 4321 + rank)
        st.set_batch(torch.from_numpy(xc).to(dev), torch.from_numpy(yc).to(dev))
        broadcast_flat(st.flat.param, world)
        st.capture()
        k = max(10, min(args.steps, 50))
        el, mm = timed_steps(st, Nc, k, min(args.warmup, 5), barrier, world, dist, dev)
        v = Bc * world * k / el
        out = {"workload": what, "value": round(v, 3), "unit": "clouds/s", "per_gpu_batch": Bc, "global_batch": Bc * world,
               "steps": k, "ms_per_step": round(1e3 * el / k, 3), "final_loss": round(mm["loss"], 5),
               "whole_step": whole_step_roofline(cfg, v / world)}
        del st, m
        torch.cuda.empty_cache()
        return out
    config_s = config_kt = None
    if not args.no_other_configs:
        config_s = other_config(CFG_S, "BASELINE.json configs[3]: 65536 pts, 13 classes, 5 encoder layers [16,64,128,256,512], bs=8 per GPU")
        config_kt = other_config(CFG_KT, "BASELINE.json configs[4] per-GPU shard: 122880 pts, 20 classes, 4 encoder layers, bs=2 per GPU")

    # secondary line: eval-mode forward clouds/s with the weights as trained so far, same batch shape
    infer = None
    if rank == 0 and not args.no_inference:
        inf = InferStep(model, B, N, use_graph=not args.no_graph)
        inf.inp.copy_(stepper.inp)
        inf.capture()
        for _ in range(args.warmup):
            inf.step(np.random.permutation(N))
        torch.cuda.synchronize(dev)
        ti = time.perf_counter()
        for _ in range(args.steps):
            inf.step(np.random.permutation(N))
        torch.cuda.synchronize(dev)
        ti = time.perf_counter() - ti
        if not bool(torch.isfinite(inf.logits).all()):
            raise SystemExit("bench.py: eval-mode logits are not finite - the measurement is invalid")
        infer = {"metric": "inference clouds/sec (eval forward), one GPU", "value": round(B * args.steps / ti, 2),
                 "ms_per_batch": round(ti / args.steps * 1e3, 3)}
        model.train()

    roof = breakdown = cpu = None
    if not args.no_roofline:
        # every rank runs the instrumented eager steps (they contain the gradient all-reduce, a collective);
        # rank 0's timings are the ones reported
        roof, breakdown = roofline_pass(stepper)
        roof.update(step_rooflines(breakdown, B, value, world))
        breakdown.pop("all_shapes", None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()             # every collective is done: the other ranks leave, rank 0 times the host path
    callers_train = callers_predict = train_py = None
    if rank == 0 and not args.no_callers:
        callers_train = trainer_e2e(dev)
        callers_predict = predict_latency()
        train_py = train_py_setting(dev)
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        is_metric_cfg = B == CFG["per_gpu_batch"]
        line = {
            # BASELINE.json's metric string only for BASELINE.json's batch; any other --batch says so in the label
            "metric": METRIC if is_metric_cfg else f"training clouds/sec, N=40960 pts, bs={B} per GPU (NOT BASELINE.json's bs=8 metric)",
            "value": round(value, 3),
            "unit": "clouds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 storage/accumulate, bf16x3 MFMA in the wide GEMMs" if WIDE_GEMM == "bf16x3" else
                     ("f32" if WIDE_GEMM == "fp32" else "f32 storage/accumulate, bf16 MFMA in the wide GEMMs"),
            "data": "synthetic",
            "storage": _o.get_storage(),
            "config": {"workload": f"RandLA-Net train step: 40960 pts/cloud, bs={B} per GPU, 2 classes, k=16, 4 encoder layers "
                                   "[16,64,128,256], dice loss + Adam", "per_gpu_batch": B,
                       "global_batch": B * world, "parallelism": f"dp{world}", "graph": not args.no_graph,
                       "wide_gemm": WIDE_GEMM},
            "final_loss": round(metrics["loss"], 5),
            "final_mIoU": round(metrics["mIoU"], 4),
            "roofline": roof,
            "cpu_baseline": cpu,
            # the path-level figures once more at the top level (the driver's parser keeps flat keys only)
            "whole_step": roof.get("whole_step") if roof else None,
            "knn": roof.get("knn") if roof else None,
            "mfma_by_level": roof.get("mfma_by_level") if roof else None,
            "config_A": config_a,
            "bf16_operands": bf16_ops,
            "fp32_exact": fp32_exact,
            "bf16_storage": bf16_storage,
            "config_S": config_s,
            "config_Kt_shard": config_kt,
            "strong_bs8": strong,
            "per_gpu_batch_sweep": sweep,
            "trainer_e2e": callers_train,
            "predict_P": callers_predict,
            "train_py_setting": train_py,
            "inference": infer,
        }
        # the side objects once more as FLAT scalars / strings (the driver's parser keeps top-level scalars only)
        flat = {
            "fp32_exact_value": fp32_exact["value"] if fp32_exact else None,
            "bf16_operands_value": bf16_ops["value"] if bf16_ops else None,
            "bf16_storage_value": bf16_storage["value"] if bf16_storage else None,
            "config_A_value": config_a["value"] if config_a else None,
            "config_S_value": config_s["value"] if config_s else None,
            "config_Kt_shard_value": config_kt["value"] if config_kt else None,
            "trainer_e2e_device_value": callers_train["rng_device"]["value"] if callers_train else None,
            "trainer_e2e_numpy_value": callers_train["rng_numpy"]["value"] if callers_train else None,
            "predict_P_ms": callers_predict["value"] if callers_predict else None,
            "train_py_setting_value": train_py["value"] if train_py else None,
            "inference_value": infer["value"] if infer else None,
            "fixed_ms_per_step": sweep.get("fixed_ms_per_step") if sweep else None,
            "marginal_ms_per_cloud": sweep.get("marginal_ms_per_cloud") if sweep else None,
            "kernel_launches_per_step": roof.get("kernel_launches_per_step") if roof else None,
            "whole_step_frac": roof["whole_step"]["frac"] if roof and roof.get("whole_step") else None,
            "roofline_frac": roof.get("frac") if roof else None,
            "roofline_kernel": roof.get("kernel") if roof else None,
            "by_function_top3": "; ".join(f"{e['kernel']} {e['ms_per_step']:.3f} ms ({e['frac']:.2f} of {e['bound']})"
                                          for e in roof["by_function"][:3]) if roof and roof.get("by_function") else None,
            "cpu_baseline_value": cpu["value"] if cpu else None,
        }
        line.update(flat)
        if breakdown is not None:
            os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
            with open(os.path.join(REPO, "gpurun_out", f"bench_breakdown_n{world}.json"), "w") as f:
                json.dump(breakdown, f, indent=1)
            print(json.dumps(breakdown), file=sys.stderr)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
