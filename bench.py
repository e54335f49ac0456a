#!/usr/bin/env python3
"""Headline benchmark: servo updates/sec for ViT-B/16 224x224 frame pairs (BASELINE.json configs[1]).

One "step" = one compute_velocity update per GPU: both frames forwarded through block 11 (I_des is
recomputed, as the reference does), dense cosine correspondence, mutual-NN filter, 24 features drawn
in a fresh random visiting order, interaction matrix, pseudo-inverse -> v_c.  Inputs (frames, depth,
intrinsics, weights, one visiting order per update) are resident in HBM before the timed region; each step is
enqueued without host synchronisation (86 stream launches), and with N > 1
every step ends with an RCCL all-gather of the 6 doubles of v_c.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision bf16|fp32] [--pairs B] [--config KEY]

Prints ONE JSON line (rank 0).  `roofline` is for the kernel class with the largest share of the step,
timed with HIP event pairs on the launch stream in a second, instrumented pass over the same steps
(`value` comes from the un-instrumented pass).  `cpu_baseline` is the CPU oracle (PyTorch-CPU fp32
forward + the reference's correspondence/control-law arithmetic) timed on this host at N = 1.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # before HIP initialises (see vit-vs_amd/__init__.py)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import vitvs_amd  # noqa: E402,F401
from vitvs_amd import _lib, config, synth, weights  # noqa: E402
from vitvs_amd import dist as vdist  # noqa: E402
from vitvs_amd.engine import Engine  # noqa: E402

PEAK_MFMA = {"bf16": 2.5e15, "fp16": 2.5e15, "fp32": 157.3e12}   # dense, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12


def split_k(m, n, k, bk):
    """Mirror of splitk_slices() in vit-vs_amd/csrc/gemm.hip."""
    tiles = -(-m // 64) * (n // 64)
    best = 1
    for c in (2, 3, 4, 6, 8):
        if k % (c * bk) == 0 and k // c >= 4 * bk and tiles * c <= 256:
            best = c
    return best


def kernel_work(cfg, n_img, n_pairs, es, binned):
    """Algorithmic FLOPs and minimum HBM bytes per LAUNCH of each kernel class."""
    n, t, d, h = cfg.seq, cfg.tokens, cfg.dim, cfg.hidden
    m = n_img * n
    bk = 128 // es
    s_proj, s_fc2 = split_k(m, d, d, bk), split_k(m, d, h, bk)
    s_avg = (s_proj + s_fc2) / 2
    dp = d * (9 if binned else 1)
    kp = -(-cfg.patch_k // 64) * 64
    s_pe = split_k(n_img * t, d, kp, bk)
    return {
        "patchify": (0.0, n_img * cfg.img_size ** 2 * 3 + n_img * t * kp * es),
        # split-K patch embedding, finished (with cls / pos_embed / block 0's norm1) by the "layernorm" class
        "patch_embed": (2.0 * n_img * t * cfg.patch_k * d, n_img * t * kp * es + d * kp * es + s_pe * n_img * t * d * 4),
        "layernorm": (10.0 * m * d, s_pe * n_img * t * d * 4 + n * d * 4 + m * d * (4 + es)),
        "qkv": (2.0 * m * 3 * d * d, (m * d + 3 * d * d + m * 3 * d) * es),
        "attention": (4.0 * n_img * n * n * d, (m * 3 * d + m * d) * es),
        "proj": (2.0 * m * d * d, (m * d + d * d) * es + s_proj * m * d * 4),
        "fc1": (2.0 * m * h * d, (m * d + h * d + m * h) * es),
        "fc2": (2.0 * m * d * h, (m * h + d * h) * es + s_fc2 * m * d * 4),
        "residual_ln": (12.0 * m * d, m * d * (4 + 4 * s_avg + 4 + es)),   # the last one also writes the descriptors
        "descriptors": (3.0 * n_img * t * dp, n_img * t * (d + dp) * 4),    # binned descriptors only
        "gram_argmax": (2.0 * n_pairs * t * t * dp, n_pairs * 2 * t * dp * 4),
        "servo": (0.0, n_pairs * t * 16),
    }


def plain_chain_us(prec, m, n, k, slices, dev, reps=300):
    """Per-launch time of the split-K GEMM in a chain of PLAIN launches (no per-launch events).  Event-stamped
    launches (the instrumented pass, rocprofv3) read ~1.7 us longer per launch than the same kernel costs in the
    un-instrumented stream (tools/launch_floor.hip), so this is the figure that adds up to `ms_per_step`."""
    import ctypes as C
    lib = _lib.load()
    code = {"bf16": _lib.BF16, "fp16": _lib.F16, "fp32": _lib.F32}[prec]
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[prec]
    a = torch.zeros((m, k), dtype=dt, device=dev)
    ws = [torch.zeros((n, k), dtype=dt, device=dev) for _ in range(12)]   # 12 weight sets, like the 12 blocks
    part = torch.zeros((slices, m, n), dtype=torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def chain(r):
        for i in range(r):
            lib.vitvs_op_linear_partial(code, C.c_void_p(a.data_ptr()), C.c_void_p(ws[i % 12].data_ptr()),
                                        C.c_void_p(part.data_ptr()), m, n, k, slices, st)
    chain(20)
    torch.cuda.synchronize(dev)
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        chain(reps)
        e1.record()
        torch.cuda.synchronize(dev)
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def cpu_baseline(cfg, sd, des, cur, depth, params, budget_s=20.0):
    from oracle import servo_ref as sr
    from oracle import vit_ref
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    frames = np.stack([des, cur])

    def update():
        toks = vit_ref.block_tokens(sd, frames, patch=cfg.patch, stride=cfg.stride, heads=cfg.heads, layer=cfg.layer,
                                    mean=cfg.mean, std=cfg.std)[:, 1:]
        return sr.servo_update(toks[0], toks[1], depth, num_pairs=params.num_pairs, input_size=cfg.img_size,
                               u_max=params.u_max, v_max=params.v_max, fx=params.f_x, fy=params.f_y,
                               lam=params.lambda_)
    torch.manual_seed(121)
    out = update()
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 20 and (time.perf_counter() < t_end or len(times) < 2):
        t0 = time.perf_counter()
        out = update()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return out, dict(value=1.0 / med, unit="updates/s", cores=torch.get_num_threads(), kind="port",
                     sample=f"{len(times)} updates of the same ViT-B/16-class frame pair (median {med * 1e3:.1f} ms, "
                            f"PyTorch-CPU fp32 forward + reference correspondence loop + numpy pinv)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--pairs", type=int, default=1, help="frame pairs per step per GPU")
    ap.add_argument("--config", default="vitb16_224", choices=sorted(config.BASELINE_CONFIGS))
    ap.add_argument("--selection", default="order", choices=["order", "dense"],
                    help="order: num_pairs features in a fresh random order (headline); dense: every mutual NN enters L_e")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plain-chain", action="store_true",
                    help="skip the plain-launch chain of the dominant GEMM (profiler runs: keeps the kernel trace to the steps)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback for the hot path)"
    # VITVS_BENCH_SHARE_GPU=1 + VITVS_DIST_BACKEND=gloo: rehearse the N > 1 control flow on a one-GPU box
    share = os.environ.get("VITVS_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # VITVS_BENCH_FORCE_DIST=1: run the N > 1 code path (process group, per-update all-gather) in a world of one rank —
    # the only way to exercise the RCCL path on a one-GPU box
    multi = world > 1 or os.environ.get("VITVS_BENCH_FORCE_DIST") == "1"
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VITVS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = config.baseline_config(args.config)
    binned = False  # north_star path: token descriptors (binning is a tested option, not the headline)
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=binned)
    sd = weights.synthetic_state_dict(cfg, 0)
    B = args.pairs
    dense = args.selection == "dense"
    eng = Engine(cfg, params, precision=args.precision, max_pairs=B, max_rows=cfg.tokens if dense else None).load_state_dict(sd)

    # per-rank synthetic inputs, resident in HBM
    seed0 = synth.ACCEPTED_FRAME_SEEDS[args.config]
    pairs = [synth.frame_pair(cfg.img_size, seed0 + 1000 * rank + i) for i in range(B)]
    des_np = np.stack([p[0] for p in pairs])
    cur_np = np.stack([p[1] for p in pairs])
    depth_np = synth.depth_pattern()
    I_des = torch.from_numpy(des_np).to(dev)
    I_cur = torch.from_numpy(cur_np).to(dev)
    Z = torch.from_numpy(np.stack([depth_np] * B)).to(dev)
    K = torch.tensor([params.intrinsics()] * B, dtype=torch.float64, device=dev)
    total = args.warmup + args.steps
    gen = torch.Generator().manual_seed(121 + rank)
    orders = torch.stack([torch.stack([torch.randperm(cfg.tokens, generator=gen) for _ in range(B)])
                          for _ in range(total)]).to(torch.int32).to(dev)       # [total, B, T]
    order_buf = torch.empty((B, cfg.tokens), dtype=torch.int32, device=dev)
    v = torch.zeros((B, 6), dtype=torch.float64, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    v_all = torch.zeros((world * B, 6), dtype=torch.float64, device=dev) if multi else None
    # N > 1: one synchronous v_c all-gather per update (torch.distributed on RCCL).  VITVS_ASYNC_GATHER=1 issues it
    # asynchronously on RCCL's own stream with alternating buffers (vit-vs_amd/dist.py: VelocityGather) — measured on one
    # GPU in a world of one rank that is SLOWER (0.577 vs 0.464 ms per update; no gather: 0.449): work of two queues
    # alternates on this platform instead of overlapping, so the second queue costs more than the wait it removes.
    async_gather = (multi and os.environ.get("VITVS_DIST_BACKEND", "nccl") == "nccl"
                    and os.environ.get("VITVS_ASYNC_GATHER") == "1")
    gather = vdist.VelocityGather(world * B, dev) if async_gather else None
    v_slots = [v, torch.zeros_like(v)]

    stream = torch.cuda.Stream(device=dev)

    def step(i):
        # a fresh visiting order per update, already resident (launches are eager, so the pointer may change)
        vi = v_slots[i & 1] if async_gather else v
        if dense:
            eng.compute_velocity_dev(I_cur, I_des, Z, K, _lib.SELECT_DENSE, None, None, False, vi, status)
        else:
            eng.compute_velocity_dev(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i], None, False, vi, status)
        if async_gather:
            gather.post(vi, i)
        elif multi:
            vdist.gather_velocities(vi, world * B, out=v_all)

    def fence():
        if gather is not None:
            gather.finish()          # the last update's all-gather belongs to the timed region
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        fence()
        elapsed = time.perf_counter() - t0
        if multi:
            te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            elapsed = float(te.item())
        status_host = status.cpu().numpy().copy()
        v_host = v_slots[(args.warmup + args.steps - 1) & 1].cpu().numpy().copy() if async_gather else v.cpu().numpy().copy()

        # single-update latency with a host synchronisation per update (a control loop's view)
        lat = []
        for i in range(10):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            step(args.warmup + i % max(args.steps, 1))
            torch.cuda.synchronize(dev)
            lat.append(time.perf_counter() - t1)

        # instrumented pass: HIP event pairs around every launch, on the launch stream
        eng.timing_enable(True)
        n_prof = min(args.steps, 50)
        for i in range(n_prof):
            step(args.warmup + i)
        if gather is not None:
            gather.finish()
        prof = eng.timing_collect()
        eng.timing_enable(False)
        plain = None
        if rank == 0 and world == 1 and not args.no_plain_chain:   # single-process extra; ranks stay in lock step at N > 1
            m_rows, bk_ = 2 * B * cfg.seq, 128 // (4 if args.precision == "fp32" else 2)
            plain = {name: plain_chain_us(args.precision, m_rows, cfg.dim, kk, split_k(m_rows, cfg.dim, kk, bk_), dev)
                     for name, kk in (("proj", cfg.dim), ("fc2", cfg.hidden))}

    updates = world * B * args.steps
    value = updates / elapsed
    es = 4 if args.precision == "fp32" else 2
    work = kernel_work(cfg, 2 * B, B, es, binned)
    overhead_s = 0.0   # the event pairs are stamped by the dispatch itself (hipExtLaunchKernelGGL): no correction
    kernels = {}
    for name, (ms, cnt) in prof.items():
        if cnt == 0:
            continue
        avg = max(ms / cnt * 1e-3 - overhead_s, 1e-7)
        fl, by = work[name]
        kernels[name] = dict(launches_per_step=cnt / n_prof, avg_us=round(avg * 1e6, 3),
                             step_share_us=round(avg * cnt / n_prof * 1e6, 2),
                             tflops=round(fl / avg / 1e12, 3), gbps=round(by / avg / 1e9, 1))
    # roofline object: the kernel SYMBOL with the largest share of the step (proj and fc2 are the same
    # split-K kernel, as rocprofv3 --stats reports them), priced with its algorithmic work per launch
    prec_tag = {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.precision]
    groups = {"linear_partial(proj+fc2)": ["proj", "fc2"]}
    for k in kernels:
        if k not in ("proj", "fc2"):
            groups[k] = [k]
    def g_share(g): return sum(kernels[c]["step_share_us"] for c in groups[g] if c in kernels)
    dom = max(groups, key=g_share)
    members = [c for c in groups[dom] if c in kernels]
    launches = sum(kernels[c]["launches_per_step"] for c in members)
    avg_s = g_share(dom) / launches * 1e-6
    fl = sum(work[c][0] * kernels[c]["launches_per_step"] for c in members) / launches
    by = sum(work[c][1] * kernels[c]["launches_per_step"] for c in members) / launches
    def narrow_tile(m, n, kk):   # mirror of the tile choice for the N = D layers in csrc/gemm.hip
        s_ = split_k(m, n, kk, 128 // es)
        if s_ == 1 and n % 128 == 0 and -(-m // 128) * (n // 128) >= 256:
            return "128,128,1"
        tiles = -(-m // 64) * (n // 64) * s_
        return "64,64,2" if tiles <= 256 and (kk // s_ // (128 // es)) >= 4 else "64,64,1"
    symbol = {"linear_partial(proj+fc2)": f"linear_kernel<{prec_tag},{narrow_tile(2 * B * cfg.seq, cfg.dim, cfg.hidden)}>:EpiPartial",
              "residual_ln": f"residual_ln_kernel<{prec_tag}>", "fc1": f"linear_kernel<{prec_tag},64,96,2>:EpiStore",
              "qkv": f"linear_kernel<{prec_tag},64,64,2>:EpiStore", "attention": "attention_f32_kernel" if prec_tag == "f32" else f"attention_16_kernel<{prec_tag}>"}.get(dom, dom)
    traffic, mfma_busy = None, None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.isfile(pmc_path) and args.precision == "bf16" and args.config == "vitb16_224" and B == 1:
        with open(pmc_path) as fh:
            pmc = json.load(fh).get(symbol, {})
        traffic = pmc.get("hbm_bytes_per_launch")
        mfma_busy = pmc.get("mfma_busy_cycles_per_launch")
    mfma_bound = dom in ("qkv", "fc1", "attention", "patch_embed", "gram_argmax", "linear_partial(proj+fc2)")
    if mfma_bound:
        peak = PEAK_MFMA["fp32" if dom == "gram_argmax" else args.precision]
        roof = dict(kernel=symbol, classes=members, bound="mfma", achieved=round(fl / avg_s / 1e12, 3), peak=peak / 1e12,
                    unit="TFLOP/s", frac=round(fl / avg_s / peak, 5), traffic=traffic,
                    algorithmic_flops_per_launch=fl, algorithmic_bytes_per_launch=by,
                    avg_launch_us=round(avg_s * 1e6, 3), event_pair_overhead_us=round(overhead_s * 1e6, 3))
        if mfma_busy:
            # rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES (separate pass, profiles/): busy cycles over the 1024 SIMDs of the chip
            roof.update(mfma_busy_cycles_per_launch=mfma_busy,
                        mfma_util_pmc=round(mfma_busy / (1024 * avg_s * 2.4e9), 5))
        if plain and dom == "linear_partial(proj+fc2)":
            # the same kernel in a chain of plain launches (what the un-instrumented step pays per launch)
            p_us = sum(plain[c] * kernels[c]["launches_per_step"] for c in members) / launches
            roof.update(plain_launch_us=round(p_us, 3), achieved_plain=round(fl / (p_us * 1e-6) / 1e12, 3),
                        frac_plain=round(fl / (p_us * 1e-6) / peak, 5))
    else:
        roof = dict(kernel=symbol, classes=members, bound="hbm", achieved=round(by / avg_s / 1e9, 2), peak=PEAK_HBM / 1e9,
                    unit="GB/s", frac=round(by / avg_s / PEAK_HBM, 5), traffic=traffic,
                    algorithmic_bytes_per_launch=by, avg_launch_us=round(avg_s * 1e6, 3),
                    event_pair_overhead_us=round(overhead_s * 1e6, 3))

    out = dict(
        metric="servo_updates_per_sec", value=round(value, 2), unit="updates/s", n_gpus=world, steps=args.steps,
        warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4), higher_is_better=True, scaling="weak",
        vs_baseline=None, dtype=args.precision, data="synthetic",
        config=dict(workload=f"{cfg.model_type} {cfg.img_size}x{cfg.img_size} frame pair(s): both frames forwarded "
                             f"through block {cfg.layer}, cosine correspondence, mutual-NN, {params.num_pairs} features "
                             f"in a fresh random order, L_e, pinv -> v_c; I_des recomputed every update",
                    key=args.config, pairs_per_step_per_gpu=B, tokens=cfg.tokens, dim=cfg.dim,
                    parallelism=(f"dp{world} (frame pairs sharded, v_c all-gather per step"
                                 f"{', asynchronous' if async_gather else ''})") if world > 1 else "single GPU",
                    weights="synthetic seed 0", selection="DENSE" if dense else "ORDER"),
        roofline=roof,
        cpu_baseline=None,
        path=dict(gflop_per_update=round(cfg.flops_per_pair(binned) / 1e9, 3),
                  tflops=round(cfg.flops_per_pair(binned) * value / world / 1e12, 3),
                  frac_of_mfma_peak=round(cfg.flops_per_pair(binned) * value / world / PEAK_MFMA[args.precision], 5),
                  weight_bytes=cfg.weight_elems() * es,
                  weight_stream_frac_of_hbm_peak=round(cfg.weight_elems() * es * value / world / B / PEAK_HBM, 5)),
        latency_ms_with_host_sync=round(float(np.median(lat)) * 1e3, 4),
        kernels=kernels,
        status=[int(s) for s in status_host],
        v_c=[float(x) for x in v_host[0]],
    )
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        ref, base = cpu_baseline(cfg, sd, des_np[0], cur_np[0], depth_np, params)
        out["cpu_baseline"] = base
        det = eng.last_details(1)
        if ref.get("corr") is not None:
            out["parity"] = dict(nn_1_agreement=float((det["nn_1"][0] == ref["corr"]["nn_1"].numpy()).mean()),
                                 nn_2_agreement=float((det["nn_2"][0] == ref["corr"]["nn_2"].numpy()).mean()),
                                 note="GPU (this dtype) vs CPU oracle argmax on the benchmarked pair; v_c parity is "
                                      "asserted by tests/test_gpu_path.py given identical selections")
    if rank == 0:
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
