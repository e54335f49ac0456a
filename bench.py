#!/usr/bin/env python3
"""Headline benchmark: servo updates/sec for ViT-B/16 224x224 frame pairs (BASELINE.json configs[1]).

One "step" = one compute_velocity update per GPU: both frames forwarded through block 11 (I_des is
recomputed, as the reference does), dense cosine correspondence, mutual-NN filter, 24 features drawn
in a fresh random visiting order, interaction matrix, pseudo-inverse -> v_c.  Inputs (frames, depth,
intrinsics, weights, one visiting order per update) are resident in HBM before the timed region; each step is
enqueued without host synchronisation (86 launches), and with N > 1
every step ends with an RCCL all-gather of the 6 doubles of v_c.

The steps of a throughput run do not depend on each other: `value` is measured with `--in-flight` (default 4) of them
enqueued at a time, each through its own handle on its own high-priority stream (vit-vs_amd/pipeline.py: one copy of the
weights, graph replay per slot, the in-flight tile plan); W warm-up steps, then exactly K timed steps between barriers and
device synchronisations, as for one stream.  The line's `sequential` object is the same W + K steps with ONE update in flight
(one handle, one stream, plain launches) — what `value` was in rounds 1-2 — measured first, on every rank.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision bf16|fp16|f16x2|fp32] [--pairs B] [--config KEY] [--in-flight D]

`--gpus N` with N > 1 and no launcher in the environment (WORLD_SIZE unset) starts the N ranks itself — one process
per GPU, before this process touches a GPU — and relays rank 0's line; it fails if fewer than N devices are visible.
Under `torch.distributed.run` (WORLD_SIZE set) each process is one rank, as the driver launches it.

Prints ONE JSON line (rank 0).  `roofline` is for the kernel symbol with the largest share of the step, timed with
HIP event pairs on the launch stream in a second, instrumented pass over the same steps, one update at a time on one of the
pipeline's handles (the kernels and the tile plan of `value`'s pass; `value` itself comes from the un-instrumented pass).  `parity` compares the benchmarked update with the CPU oracle under the same visiting order;
`secondary` is the fp32 parity mode measured in the same process; `cpu_baseline` is the CPU oracle (PyTorch-CPU fp32
forward + the reference's correspondence/control-law arithmetic) timed on this host at N = 1.
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # before HIP initialises (see vit-vs_amd/__init__.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense, MI355X_MICROARCH.md.  f16x2 (split-f16: three f16 MFMAs per k-step for fp32-class results) is priced against the SAME f16
# peak on its algorithmic FLOPs — a third of the peak is the most the scheme can reach, said beside every fraction it reports
PEAK_MFMA = {"bf16": 2.5e15, "fp16": 2.5e15, "f16x2": 2.5e15, "fp32": 157.3e12}
ELEM_BYTES = {"bf16": 2, "fp16": 2, "f16x2": 4, "fp32": 4}       # bytes per logical operand element (f16x2: an fp16 hi / lo pair)
PEAK_HBM = 8.0e12
# separate rocprofv3 --pmc passes / rocprofv3 --kernel-trace --stats of the same command (tools/measure_round.sh), per precision;
# the newest committed round that has the file is quoted
def _profile(name):
    for r in ("r05", "r04"):
        if os.path.isfile(os.path.join(ROOT, "profiles", f"{r}_{name}")):
            return os.path.join("profiles", f"{r}_{name}")
    return os.path.join("profiles", f"r05_{name}")
TRAFFIC_PROFILES = {"bf16": "pmc_traffic.json", "f16x2": "pmc_traffic_f16x2.json"}
STATS_PROFILES = {"bf16": "kernel_stats_bf16.csv", "f16x2": "kernel_stats_f16x2.csv"}


# ----------------------------------------------------------------------------------------------- launcher
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "f16x2", "fp32"],
                    help="operand type of the forward; f16x2 = split-f16 (fp16 hi / lo pairs, fp32-class results: the parity mode at servo rate)")
    ap.add_argument("--pairs", type=int, default=1, help="frame pairs per step per GPU")
    ap.add_argument("--config", default="vitb16_224")
    ap.add_argument("--selection", default="order", choices=["order", "dense"],
                    help="order: num_pairs features in a fresh random order (headline); dense: every mutual NN enters L_e")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="independent updates in flight per GPU (handles x streams, vit-vs_amd/pipeline.py); 0 = the measured "
                         "default 4; 1 = one stream, as in rounds 1-2")
    ap.add_argument("--binned", action="store_true",
                    help="3x3 log-binned descriptors (use_feature_binning: true, the reference's shipped default with "
                         "--config vits14_308: config.yaml:17, vitvs_v2.py:482-493); the Gram's K becomes 9 D")
    ap.add_argument("--no-gather", action="store_true",
                    help="N > 1: skip the per-update v_c all-gather (every rank keeps its own twists: SURVEY 8(e) allows it); "
                         "with and without it the driver can separate the collective's cost from the compute")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fp32 parity-mode leg")
    ap.add_argument("--no-plain-chain", action="store_true",
                    help="skip the plain-launch chain of the dominant GEMM (profiler runs: keeps the kernel trace to the steps)")
    return ap.parse_args(argv)


def _listed(var):
    """Entries of a *_VISIBLE_DEVICES variable (None when unset; an empty string hides every device)."""
    val = os.environ.get(var)
    if val is None:
        return None
    return [x for x in val.split(",") if x.strip() != ""]


def visible_devices(sysfs="/sys/class/kfd/kfd/topology/nodes") -> int:
    """GPUs this process's children could use, counted WITHOUT loading torch or HIP in this process (a parent that has
    initialised the GPU must not start the ranks: the pool forbids exec from such a process).  KFD topology nodes with
    SIMDs are GPUs (CPU nodes have simd_count 0); ROCR_/HIP_/CUDA_VISIBLE_DEVICES narrow the count the way the runtime
    applies them.  If the topology is not readable, ask a throwaway child (started before anything else, it exits before
    any rank starts)."""
    n = None
    try:
        n = 0
        for node in sorted(os.listdir(sysfs)):
            try:
                with open(os.path.join(sysfs, node, "properties")) as fh:
                    props = dict(ln.split(None, 1) for ln in fh if " " in ln)
            except OSError:
                continue      # a node of another container's cgroup: not ours
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        n = None
    if n is None:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                           text=True, timeout=300)
        try:
            n = int(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            n = 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        lst = _listed(var)
        if lst is not None:
            n = min(n, len(lst))
    return n


def rank_environments(n_ranks: int, port: int, base_env=None):
    """Environment of each child rank (what torch.distributed.run would set)."""
    envs = []
    for r in range(n_ranks):
        e = dict(os.environ if base_env is None else base_env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=e.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        envs.append(e)
    return envs


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv, script=None, deadline_s=None, poll_s=0.2) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU), watch ALL of them, relay rank 0's JSON
    line.  This process never touches a GPU (no torch import, no HIP call: `visible_devices`), so starting children from it
    is neither an exec after GPU initialisation nor a fork of a GPU-initialised process.  The first rank that exits
    non-zero ends the run within seconds: the others are terminated (then killed), its rank and the tail of its stderr
    are printed, and the exit code is 1 — a dead rank never leaves the others waiting in a collective until a time limit
    kills the whole job without a cause on record.  An overall deadline (VITVS_BENCH_DEADLINE_S, default 900 s) bounds
    the run the same way."""
    import signal
    import tempfile
    n = args.gpus
    share = os.environ.get("VITVS_BENCH_SHARE_GPU") == "1"     # rehearsal: N ranks on fewer GPUs (gloo)
    have = visible_devices()
    if have < n and not share:
        print(f"bench.py: --gpus {n} but only {have} HIP device(s) visible; refusing to print a mislabelled line "
              f"(rehearse the N-rank control flow on fewer GPUs with VITVS_BENCH_SHARE_GPU=1 VITVS_DIST_BACKEND=gloo)",
              file=sys.stderr)
        return 2
    if have < 1:
        print("bench.py: no HIP device visible", file=sys.stderr)
        return 2
    if deadline_s is None:
        deadline_s = float(os.environ.get("VITVS_BENCH_DEADLINE_S", "900"))
    envs = rank_environments(n, free_port())
    logdir = tempfile.mkdtemp(prefix="vitvs_bench_")
    procs, outs, errs = [], [], []

    def tail(fh, lines=15):
        fh.flush()
        fh.seek(0)
        return "".join(fh.readlines()[-lines:])

    watched = (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)

    class _masked:
        """The launcher's signals held back (pthread_sigmask) for a few lines that must not be torn: a rank's start + its entry in
        `procs` (a signal between the two would leave a rank nobody ends: it runs in a session of its own), and `stop_all`."""
        def __enter__(self):
            try:
                self.old = signal.pthread_sigmask(signal.SIG_BLOCK, watched)
            except (ValueError, OSError, AttributeError):
                self.old = None
        def __exit__(self, *exc):
            if self.old is not None:
                signal.pthread_sigmask(signal.SIG_SETMASK, self.old)     # a signal that arrived meanwhile is delivered here
            return False

    stopped = []

    def stop_all():
        """Every rank runs in a session of its own (so that a signal aimed at this launcher's group cannot take a rank down
        half-way through a collective while the others wait): ending them is therefore this process's job on EVERY way out —
        a failed rank, the deadline, SIGTERM / SIGINT to the launcher, an exception.  SIGTERM to each rank's process group,
        SIGKILL after 5 s.  Idempotent; the caller holds the launcher's signals back while it runs."""
        if stopped:
            return
        stopped.append(True)
        for p_ in procs:
            if p_.poll() is None:
                try:
                    os.killpg(p_.pid, signal.SIGTERM)      # pid == pgid == sid of the rank (start_new_session)
                except (ProcessLookupError, PermissionError):
                    p_.terminate()
        t_end = time.monotonic() + 5.0
        for p_ in procs:
            try:
                p_.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p_.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    p_.kill()
                p_.wait()

    class _Signalled(Exception):
        pass

    def on_signal(signum, _frame):
        raise _Signalled(signum)

    previous = {}
    for sig in watched:
        try:
            previous[sig] = signal.signal(sig, on_signal)
        except (ValueError, OSError):                      # not the main thread (the unit tests call this function directly)
            pass
    rc, keep_logs = 1, True
    try:
        for r, env in enumerate(envs):
            outs.append(open(os.path.join(logdir, f"rank{r}.out"), "w+"))
            errs.append(open(os.path.join(logdir, f"rank{r}.err"), "w+"))
            with _masked():                                 # started and on record, or not started: never in between
                procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                              stdout=outs[r], stderr=errs[r], text=True, start_new_session=True))
        t_stop = time.monotonic() + deadline_s
        failed = None
        while True:
            codes = [p_.poll() for p_ in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > t_stop:
                failed = -1
                break
            time.sleep(poll_s)
        if failed is not None:
            stop_all()
            if failed < 0:
                print(f"bench.py: no result after {deadline_s:.0f} s; ranks terminated. rank 0 stderr tail:\n{tail(errs[0])}", file=sys.stderr)
            else:
                print(f"bench.py: rank {failed} exited with code {procs[failed].returncode}; the other ranks were terminated. "
                      f"Its stderr tail:\n{tail(errs[failed])}", file=sys.stderr)
            return 1
        out0 = tail(outs[0], lines=50)
        for r in range(n):
            t_ = tail(errs[r], lines=5)
            if t_.strip() and os.environ.get("VITVS_BENCH_VERBOSE") == "1":
                print(f"[rank {r} stderr] {t_}", file=sys.stderr)
        line = [ln for ln in out0.splitlines() if ln.startswith("{")]
        if not line:
            print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
            return 1
        parsed = json.loads(line[-1])
        if parsed.get("n_gpus") != n:
            print(f"bench.py: rank 0 reported n_gpus={parsed.get('n_gpus')} for --gpus {n}", file=sys.stderr)
            return 1
        print(line[-1])
        rc, keep_logs = 0, False
        return 0
    except _Signalled as sig:
        print(f"bench.py: signal {sig.args[0]} received; terminating the ranks", file=sys.stderr)
        rc = 128 + int(sig.args[0])
        return rc
    finally:
        with _masked():
            stop_all()                                      # no rank outlives the launcher, whichever way it leaves
        for fh in outs + errs:
            try:
                fh.close()
            except OSError:
                pass
        if not keep_logs:
            import shutil
            shutil.rmtree(logdir, ignore_errors=True)
        elif rc != 0:
            print(f"bench.py: per-rank logs kept in {logdir}", file=sys.stderr)
        for sig, h_ in previous.items():
            try:
                signal.signal(sig, h_)
            except (ValueError, OSError):
                pass


# ----------------------------------------------------------------------------------------------- work model
def split_k(m, n, k, bk, in_flight=1, big_tiles=None):
    """Mirror of splitk_slices() in vit-vs_amd/csrc/gemm.hip (bk = logical k per k-tile: 64 for the 16-bit precisions, 32 for fp32
    and f16x2; big_tiles = the precision has the 256-row tiles of gemm_big.hip: everything but fp32)."""
    if big_tiles is None:
        big_tiles = bk == 64
    if big_tiles and m >= 1024 and n % 128 == 0:
        t128 = -(-m // 256) * (n // 128)
        if t128 >= 96:
            return 1
        pick = 1
        for c in (2, 3, 4):
            if k % (c * 64) == 0 and k // c >= 8 * 64 and t128 * c <= 256:
                pick = c
        if in_flight >= 2 and pick > 2 and m >= 2048:
            pick = 2
        if pick > 1:
            return pick
    tiles = -(-m // 64) * (n // 64)
    best = 1
    for c in (2, 3, 4, 6, 8):
        if k % (c * bk) == 0 and k // c >= 4 * bk and tiles * c <= 256:
            best = c
    return min(best, 2) if in_flight >= 2 else best     # beside other queues' launches: at most two slices


def kernel_work(cfg, n_img, n_pairs, es, binned, in_flight=1, big_tiles=None):
    """Algorithmic FLOPs and minimum HBM bytes per LAUNCH of each kernel class."""
    n, t, d, h = cfg.seq, cfg.tokens, cfg.dim, cfg.hidden
    m = n_img * n
    bk = 128 // es
    s_proj, s_fc2 = split_k(m, d, d, bk, in_flight, big_tiles), split_k(m, d, h, bk, in_flight, big_tiles)
    s_avg = (s_proj + s_fc2) / 2
    dp = d * (9 if binned else 1)
    kp = -(-cfg.patch_k // 64) * 64
    s_pe = split_k(n_img * t, d, kp, bk, in_flight, big_tiles)
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
        # binned descriptors: the 9 D-wide Gram is taken as a 3 x 3 stencil over the raw D-wide token Gram (correspond.hip), so the
        # kernels' own work is: token norms, a D-wide Gram written out as T x T floats, nine adds per similarity
        "descriptors": (2.0 * n_img * t * d, n_img * t * d * 4),            # binned only: the tokens' squared norms
        "gram_argmax": ((2.0 * n_pairs * t * t * d, n_pairs * (2 * t * d + t * t) * 4) if binned else
                        (2.0 * n_pairs * t * t * dp, n_pairs * 2 * t * dp * 4)),
        "gram_stencil": (11.0 * n_pairs * t * t, n_pairs * (t * t + 4 * t) * 4),
        "servo": (0.0, n_pairs * t * 16),
    }


def plain_chain_us(prec, m, n, k, slices, dev, reps=300):
    """Per-launch time of the split-K GEMM in a chain of PLAIN launches (no per-launch events).  Event-stamped
    launches (the instrumented pass, rocprofv3) read ~1.7 us longer per launch than the same kernel costs in the
    un-instrumented stream (tools/launch_floor.hip), so this is the figure that adds up to `ms_per_step`."""
    import ctypes as C
    import torch
    from vitvs_amd import _lib
    lib = _lib.load()
    code = {"bf16": _lib.BF16, "fp16": _lib.F16, "fp32": _lib.F32, "f16x2": _lib.F16X2}[prec]
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "f16x2": torch.float16}[prec]
    kw = 2 * k if prec == "f16x2" else k                                  # f16x2 rows: an fp16 hi / lo pair per element
    a = torch.zeros((m, kw), dtype=dt, device=dev)
    ws = [torch.zeros((n, kw), dtype=dt, device=dev) for _ in range(12)]   # 12 weight sets, like the 12 blocks
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


def overlapped_chain_us(prec, m, d, hidden, s_proj, s_fc2, dev, queues=3, blocks=12, reps=6, rounds=8):
    """Per-launch time of the dominant GEMM symbol (the partial-sum kernel of proj and fc2) with `queues` queues busy — the regime
    `value` is measured in, where a kernel's own duration is not defined (kernels of several queues share the chip) and the
    instrumented pass therefore times it alone.  Each queue replays a captured chain of `reps` x `blocks` x [proj, fc2] launches
    in forward order (12 weight sets, shared by the queues like the pipeline's one copy of the weights; activations and partial
    sums per queue) on a high-priority stream of its own; the figure is wall time / launches over all queues (what
    tools/op_chain `queues` reports), for `queues` queues and for one."""
    import ctypes as C
    import torch
    from vitvs_amd import _lib
    lib = _lib.load()
    code = {"bf16": _lib.BF16, "fp16": _lib.F16, "fp32": _lib.F32, "f16x2": _lib.F16X2}[prec]
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "f16x2": torch.float16}[prec]
    g = torch.Generator(device="cpu").manual_seed(3)
    wide = 2 if prec == "f16x2" else 1                                   # f16x2 rows: an fp16 hi / lo pair per element
    rnd = lambda r, c: (torch.randn((r, c * wide), generator=g) * 0.05).to(dt).to(dev)   # noqa: E731  (random data: zeros run at a higher clock)
    w_proj = [rnd(d, d) for _ in range(blocks)]
    w_fc2 = [rnd(d, hidden) for _ in range(blocks)]
    per_queue = []
    for _ in range(queues):
        per_queue.append(dict(a=rnd(m, d), h=rnd(m, hidden), part=torch.zeros((max(s_proj, s_fc2), m, d), dtype=torch.float32, device=dev),
                              stream=torch.cuda.Stream(device=dev, priority=-1)))

    def chain(q):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for _ in range(reps):
            for i in range(blocks):
                lib.vitvs_op_linear_partial(code, C.c_void_p(q["a"].data_ptr()), C.c_void_p(w_proj[i].data_ptr()),
                                            C.c_void_p(q["part"].data_ptr()), m, d, d, s_proj, st)
                lib.vitvs_op_linear_partial(code, C.c_void_p(q["h"].data_ptr()), C.c_void_p(w_fc2[i].data_ptr()),
                                            C.c_void_p(q["part"].data_ptr()), m, d, hidden, s_fc2, st)
    for q in per_queue:
        with torch.cuda.stream(q["stream"]):
            chain(q)                                  # warm: LDS opt-in, code objects
        q["stream"].synchronize()
        q["graph"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(q["graph"], stream=q["stream"]):
            chain(q)
    launches = 2 * blocks * reps

    def timed(n_q):
        best = float("inf")
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _r in range(rounds):
                for q in per_queue[:n_q]:
                    with torch.cuda.stream(q["stream"]):
                        q["graph"].replay()
            for q in per_queue[:n_q]:
                q["stream"].synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e6 / (rounds * launches * n_q))
        return best
    timed(queues)
    return dict(queues=queues, per_launch_us=round(timed(queues), 3), per_launch_us_one_queue=round(timed(1), 3),
                launches_per_queue_and_round=launches, rounds=rounds)


# ----------------------------------------------------------------------------------------------- CPU oracle legs
def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_update(cfg, sd, des, cur, depth, params, order, exact_order=False):
    """One update by the CPU oracle under a given visiting order: tokens -> tables -> the first num_pairs mutual NNs
    met in `order` -> the reference's law.  Returns dict(nn_1, nn_2, selected, v_c, status).  `exact_order` evaluates the
    similarity matrix the way the reference does (a Python loop over the T tokens, vitvs_v2.py:49-56) instead of one matmul."""
    import numpy as np
    import torch
    from oracle import servo_ref as sr
    from oracle import vit_ref
    toks = vit_ref.block_tokens(sd, np.stack([des, cur]), patch=cfg.patch, stride=cfg.stride, heads=cfg.heads,
                                layer=cfg.layer, mean=cfg.mean, std=cfg.std)[:, 1:]
    sim = sr.cosine_matrix(toks[0], toks[1], exact_order=exact_order)
    _, nn1, _, nn2 = sr.nearest_neighbours(sim)
    nn1, nn2 = nn1.numpy(), nn2.numpy()
    t, g, k = cfg.tokens, cfg.grid, params.num_pairs
    mutual = nn2[nn1] == np.arange(t)
    out = dict(nn_1=nn1, nn_2=nn2, sim=sim)
    if mutual.all() or not mutual.any():
        out.update(status=1, selected=np.zeros(0, np.int64), v_c=np.zeros(6))
        return out
    sel = np.array([i for i in np.asarray(order, dtype=np.int64) if mutual[i]][:k], dtype=np.int64)
    out["selected"] = sel
    out.update(oracle_law(cfg, params, sel, nn1[sel], depth))
    return out


def oracle_law(cfg, params, sel, matches, depth):
    import numpy as np
    import torch
    from oracle import servo_ref as sr
    g = cfg.grid
    p1 = torch.from_numpy(np.stack([sel // g, sel % g], 1).astype(np.int64))
    p2 = torch.from_numpy(np.stack([matches // g, matches % g], 1).astype(np.int64))
    s_star, s = sr.calculate_uv(sr.patch_centres(p1, cfg.img_size, g), sr.patch_centres(p2, cfg.img_size, g),
                                params.num_pairs, params.u_max, params.v_max, cfg.img_size)
    res = sr.velocity(s_star, s, depth, params.f_x, params.f_y, params.c_x, params.c_y, params.lambda_)
    return dict(v_c=res["v_c"], status=0 if (len(sel) >= 4 or len(sel) == params.num_pairs) else 2)


def cpu_baseline(cfg, sd, des, cur, depth, params, budget_s=12.0):
    """The CPU oracle timed on this host: on the threads this process may use (its CPU affinity — a GPU
    box gives one GPU's share of the host), on os.cpu_count() threads (every core of the node, as §8(d) words it;
    oversubscribed when the affinity is narrower), on one thread, and in the reference's own shape — the similarity
    matrix by a Python loop over the T tokens (vitvs_v2.py:49-56) instead of one matmul.  `value` is the best of the
    multi-thread figures with the matmul form; every figure is in the object."""
    import numpy as np
    import torch
    order = np.arange(cfg.tokens)
    usable = max(1, len(os.sched_getaffinity(0)))
    host = os.cpu_count() or usable

    def timed(n_threads, budget, at_most, exact=False):
        torch.set_num_threads(n_threads)
        oracle_update(cfg, sd, des, cur, depth, params, order, exact)     # warm-up
        times, t_end = [], time.perf_counter() + budget
        while len(times) < at_most and (time.perf_counter() < t_end or len(times) < 2):
            t0 = time.perf_counter()
            oracle_update(cfg, sd, des, cur, depth, params, order, exact)
            times.append(time.perf_counter() - t0)
        return times
    def cpu_quota():
        """CPUs the cgroup lets this process use at once (a GPU box hands out one GPU's share of the host whatever the
        affinity mask says); None when unlimited or unknown."""
        for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            try:
                with open(path) as fh:
                    parts = fh.read().split()
                if path.endswith("cpu.max"):
                    return None if parts[0] == "max" else max(1, int(int(parts[0]) / int(parts[1])))
                q = int(parts[0])
                if q <= 0:
                    return None
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    return max(1, int(q / int(fh.read().split()[0])))
            except (OSError, ValueError, IndexError):
                continue
        return None
    quota = cpu_quota()
    legs, skipped = {}, {}
    base = min(16, usable)
    legs[base] = timed(base, budget_s, 20)
    base_med = float(np.median(legs[base]))
    for n_thr in sorted({usable, host} - {base}):
        if quota is not None and n_thr > 2 * quota:
            skipped[str(n_thr)] = f"not timed: the cgroup allows {quota} CPUs, {n_thr} threads would only time-slice them"
            continue
        # a cheap guard first: an oversubscribed pool (more threads than CPUs really granted) makes one update take tens of
        # seconds; a few small matmuls show it in well under a second
        def probe(threads):
            torch.set_num_threads(threads)
            a_ = torch.ones((768, 768))
            torch.mm(a_, a_)
            t0 = time.perf_counter()
            for _ in range(8):
                torch.mm(a_, a_)
            return time.perf_counter() - t0
        p_base, p_thr = probe(base), probe(n_thr)
        if p_thr > 3.0 * p_base:
            skipped[str(n_thr)] = (f"not timed: 8 matmuls of 768^3 take {p_thr * 1e3:.0f} ms on {n_thr} threads against "
                                   f"{p_base * 1e3:.0f} ms on {base}: the pool is oversubscribed")
            continue
        legs[n_thr] = timed(n_thr, 4.0, 20)
    best_thr = min(legs, key=lambda k_: float(np.median(legs[k_])))
    many = legs[best_thr]
    loop = timed(best_thr, 4.0, 10, exact=True)
    one = timed(1, 4.0, 5)
    torch.set_num_threads(min(16, usable))
    med, p90, med1 = float(np.median(many)), float(np.percentile(many, 90)), float(np.median(one))
    medl = float(np.median(loop))
    return dict(value=round(1.0 / med, 3), unit="updates/s", cores=best_thr, kind="port",
                p90_ms=round(p90 * 1e3, 2), median_ms=round(med * 1e3, 2), threads_1_value=round(1.0 / med1, 3),
                threads_1_median_ms=round(med1 * 1e3, 2), cpu_model=cpu_model(), host_cpus=host, usable_cpus=usable,
                cgroup_cpus=quota, thread_counts_not_timed=skipped,
                by_threads={str(k_): dict(value=round(1.0 / float(np.median(v_)), 3), median_ms=round(float(np.median(v_)) * 1e3, 2),
                                          updates=len(v_)) for k_, v_ in legs.items()},
                reference_loop=dict(value=round(1.0 / medl, 3), median_ms=round(medl * 1e3, 2), threads=best_thr, updates=len(loop),
                                    note="the reference's chunk_cosine_sim: a Python loop over the T tokens (vitvs_v2.py:49-56); "
                                         "same forward and law"),
                sample=f"{len(many)} updates of the same {cfg.model_type} {cfg.img_size}x{cfg.img_size} frame pair on {best_thr} "
                       f"threads (median {med * 1e3:.1f} ms, p90 {p90 * 1e3:.1f} ms; tried {sorted(legs)} threads, {usable} usable "
                       f"of {host} host CPUs), {len(loop)} with the reference's per-token similarity loop (median {medl * 1e3:.1f} "
                       f"ms) and {len(one)} on 1 thread (median {med1 * 1e3:.1f} ms): PyTorch-CPU fp32 forward + reference "
                       f"correspondence and law (numpy pinv)")


def parity_block(eng, cfg, sd, params, des, cur, depth_np, I_cur, I_des, Z, K, order_row, _lib):
    """Device vs CPU oracle on ONE update under the same visiting order."""
    import numpy as np
    v, st = eng.compute_velocity_dev(I_cur[:1], I_des[:1], Z[:1], K[:1], _lib.SELECT_ORDER, order_row[None].contiguous())
    det = eng.last_details(1)
    v = v.cpu().numpy()[0]
    order = order_row.cpu().numpy()
    ref = oracle_update(cfg, sd, des, cur, depth_np, params, order)
    k = params.num_pairs
    dev_sel = det["selected"][0, :k].astype(np.int64)
    dev_sel = dev_sel[dev_sel >= 0]
    out = dict(nn_1_agreement=float((det["nn_1"][0] == ref["nn_1"]).mean()),
               nn_2_agreement=float((det["nn_2"][0] == ref["nn_2"]).mean()),
               device_status=int(st[0]), oracle_status=int(ref["status"]))
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))   # noqa: E731
    if ref["status"] == 0 and int(st[0]) == 0:
        same_sel = bool(np.array_equal(dev_sel, ref["selected"]))
        out["selected_tokens_agree"] = same_sel
        out["selected_nn1_agree"] = float((det["nn_1"][0][ref["selected"]] == ref["nn_1"][ref["selected"]]).mean())
        # end to end: identical when the same tokens were drawn and their matches agree; otherwise another (valid) draw
        out["v_c_rel_l2"] = rel(v, ref["v_c"])
        # the law itself: the oracle's law on the DEVICE's draw and matches
        law = oracle_law(cfg, params, dev_sel, det["nn_1"][0].astype(np.int64)[dev_sel], depth_np)
        out["v_c_rel_l2_given_device_selection"] = rel(v, law["v_c"])
        out["note"] = ("v_c is a function of integer pixel features: v_c_rel_l2 is fp64 round-off when the device drew the "
                       "oracle's tokens with the oracle's matches (selected_tokens_agree and selected_nn1_agree == 1), "
                       "otherwise it compares two different valid draws")
    return out


# ----------------------------------------------------------------------------------------------- one rank
def timed_updates(eng, step, fence, warmup, steps, dev):
    import torch
    for i in range(warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    fence()
    return time.perf_counter() - t0


def run_rank(args):
    import numpy as np
    import torch
    import vitvs_amd  # noqa: F401
    from vitvs_amd import _lib, config, synth, weights
    from vitvs_amd import dist as vdist
    from vitvs_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, launched by torch.distributed.run or "
                         f"by `python bench.py --gpus N` itself")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback for the hot path)"
    # VITVS_BENCH_SHARE_GPU=1 + VITVS_DIST_BACKEND=gloo: rehearse the N > 1 control flow on a one-GPU box
    share = os.environ.get("VITVS_BENCH_SHARE_GPU") == "1"
    if local_rank >= torch.cuda.device_count() and not share:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
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
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime
        backend = os.environ.get("VITVS_DIST_BACKEND", "nccl")
        # a rank that never arrives fails the rendezvous (and every later collective) after 120 s, not after the default 10 min
        tmo = datetime.timedelta(seconds=float(os.environ.get("VITVS_DIST_TIMEOUT_S", "120")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    cfg = config.baseline_config(args.config)
    binned = bool(args.binned)  # default: the north_star path's token descriptors; --binned: the reference's shipped default
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=binned)
    sd = weights.synthetic_state_dict(cfg, 0)
    B = args.pairs
    dense = args.selection == "dense"
    eng = Engine(cfg, params, precision=args.precision, max_pairs=B, max_rows=cfg.tokens if dense else None).load_state_dict(sd)
    # Updates of a throughput run do not depend on each other: `value` is measured with `in_flight` of them enqueued on as
    # many streams through as many handles (vit-vs_amd/pipeline.py); the one-stream figure of the earlier rounds is reported
    # beside it as `sequential`.
    # default depth 4 = the hardware queues HIP gives a priority class (one slot per queue; a fifth stream shares one and the rate
    # falls 17 %).  Rounds 3-4 ran three: the empty-launch micro-benchmark overlaps the launch floor of up to three queues
    # (tools/launch_floor) and a fourth update had measured +2 %.  Round 5 swept it in the driver's own form (--steps 20 --warmup 5,
    # three runs each, one box: 3840-3909 with three, 3994-4050 with four) and over the other configurations (2 / 4 pairs +4 %,
    # ViT-S +3-4 %, f16x2 +4 %, ViT-B/8 +1 %, ViT-L/14 and 8 pairs +-1 %): profiles/r05_driver_form_depth_sweep.txt,
    # r05_depth3_vs_4.txt
    # ... while nothing else keeps a queue busy.  With N > 1 every update is followed by an all-gather: torch >= 2.8 launches a
    # synchronous collective on the CURRENT stream (this image's torch 2.10 says so itself: "TORCH_NCCL_AVOID_RECORD_STREAMS is the
    # default now"), i.e. on the slot's own queue, not on a communicator stream — in a world of one rank the kernel trace of this
    # command shows the four slot queues and no fifth (profiles/r05_notes.md section 10), and that line gains like the others
    # (4149 -> 4612).  On an older torch the communicator's stream would be a fifth queue; no multi-GPU box was available to check, so
    # the multi-GPU line calibrates before its warm-up (four slots against three, untimed; `slots_calibration` in the line).
    in_flight = args.in_flight if args.in_flight > 0 else 4
    pipe = None
    if in_flight > 1:
        from vitvs_amd.pipeline import UpdatePipeline
        pipe = UpdatePipeline(cfg, params, sd, precision=args.precision, depth=in_flight, max_pairs=B,
                              max_rows=cfg.tokens if dense else None, device=dev,
                              stream_priority=int(os.environ.get("VITVS_PIPE_PRIORITY", "-1")),
                              share_weights=os.environ.get("VITVS_PIPE_SHARE", "1") == "1",    # (A/B tooling: tools/ab_bench.sh)
                              plan_hint=os.environ.get("VITVS_PIPE_HINT", "1") == "1")

    # per-rank synthetic inputs, resident in HBM.  The headline configuration draws its pairs from the accepted rig
    # seeds (rank r, pair i -> seed index (r * B + i) mod 8), so the 8-GPU run IS configs[3]'s 8-camera rig.
    if args.config == "vitb16_224":
        seeds = [synth.RIG8_FRAME_SEEDS[(rank * B + i) % len(synth.RIG8_FRAME_SEEDS)] for i in range(B)]
    else:
        seeds = [synth.ACCEPTED_FRAME_SEEDS[args.config] + 1000 * rank + i for i in range(B)]
    pairs = [synth.frame_pair(cfg.img_size, s) for s in seeds]
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
    v = torch.zeros((B, 6), dtype=torch.float64, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    v_all = torch.zeros((world * B, 6), dtype=torch.float64, device=dev) if multi else None
    # N > 1: one v_c all-gather per update (torch.distributed on RCCL), enqueued behind the update on the update's own stream.
    # VITVS_ASYNC_GATHER=1 (one update in flight only) issues it asynchronously on RCCL's stream with alternating buffers
    # (vit-vs_amd/dist.py: VelocityGather) — measured SLOWER on one GPU in a world of one rank (0.577 vs 0.464 ms per update).
    do_gather = multi and not args.no_gather
    async_gather = (do_gather and os.environ.get("VITVS_DIST_BACKEND", "nccl") == "nccl"
                    and os.environ.get("VITVS_ASYNC_GATHER") == "1" and pipe is None)
    gather = vdist.VelocityGather(world * B, dev) if async_gather else None
    v_slots = [v, torch.zeros_like(v)]

    stream = torch.cuda.Stream(device=dev)

    def make_step(engine, goal=None):
        goal = I_des if goal is None else None          # goal = "cached": the engine holds the goal (Engine.set_goal)
        def step(i):
            # a fresh visiting order per update, already resident (launches are eager, so the pointer may change)
            vi = v_slots[i & 1] if async_gather else v
            if dense:
                engine.compute_velocity_dev(I_cur, goal, Z, K, _lib.SELECT_DENSE, None, None, False, vi, status)
            else:
                engine.compute_velocity_dev(I_cur, goal, Z, K, _lib.SELECT_ORDER, orders[i % total], None, False, vi, status)
            if async_gather:
                gather.post(vi, i)
            elif do_gather:
                vdist.gather_velocities(vi, world * B, out=v_all)
        return step
    step = make_step(eng)

    def fence():
        if gather is not None:
            gather.finish()          # the last update's all-gather belongs to the timed region
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    v_all_slots = [torch.zeros((world * B, 6), dtype=torch.float64, device=dev) for _ in range(in_flight)] if (multi and pipe) else None

    def pipe_step(i):
        # inputs_ready: the frames, depth image, intrinsics and visiting orders are device-resident and complete since before the fence
        # that opens the timed region (the contract's "inputs already resident in HBM").  Without it every submission records an event
        # on THIS stream for the slot to wait on — traffic on a fifth hardware queue beside the slots' four, which the chip time-slices:
        # same box, 300 steps 4315-4367 -> 4684-4769 updates/s, the driver's 20-step form 3951-4052 -> 4205-4323
        # (profiles/r05_ab_inputs_ready.txt)
        if dense:
            t = pipe.submit(I_cur, I_des, Z, K, _lib.SELECT_DENSE, None, None, False, inputs_ready=True)
        else:
            t = pipe.submit(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % total], None, False, inputs_ready=True)
        if do_gather:                               # the update's v_c all-gather, behind it on its own stream
            v_k, _, st_k = pipe.slot(t)
            with torch.cuda.stream(st_k):
                vdist.gather_velocities(v_k, world * B, out=v_all_slots[t % pipe.active])

    def pipe_fence():
        pipe.synchronize()
        fence()

    def rank_spread(el):
        """ms per step of this rank's timed region, reduced over the ranks: the job's figure is the max (the contract), the
        min beside it shows whether one rank lags the others."""
        t = torch.tensor([el / args.steps * 1e3], dtype=torch.float64, device=dev)
        lo, hi, tot = t.clone(), t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        return dict(min=round(float(lo.item()), 4), max=float(hi.item()), mean=round(float(tot.item()) / world, 4))

    sequential = None
    slots_calibration = None
    with torch.cuda.stream(stream):
        if pipe is not None:
            # first the same updates one at a time on one stream — rounds 1-2's `value`, same W and K, with its per-update
            # all-gather at N > 1 — then the pipelined leg that `value` reports (every rank runs both, in this order)
            el_seq = timed_updates(eng, step, fence, args.warmup, args.steps, dev)
            seq_rank_ms = None
            if multi:
                seq_rank_ms = rank_spread(el_seq)
                el_seq = seq_rank_ms["max"] * args.steps * 1e-3
            sequential = dict(metric="servo_updates_per_sec", value=round(world * B * args.steps / el_seq, 2), unit="updates/s",
                              steps=args.steps, warmup=args.warmup, ms_per_step=round(el_seq / args.steps * 1e3, 4),
                              note="one update in flight per GPU: one handle, one stream, plain launches, the one-stream tile "
                                   "plan (what `value` was in rounds 1-2); measured before the pipelined leg")
            if seq_rank_ms:
                sequential["per_rank_ms_per_step"] = dict(seq_rank_ms, max=round(seq_rank_ms["max"], 4))
            for i in range(2 * in_flight):          # set-up, not warm-up: every slot captures its graph (first call) and
                pipe_step(i)                        # uploads the instantiated graph (first replay) before anything is timed
            pipe_fence()
            if world > 1 and do_gather and args.in_flight == 0 and in_flight == 4:
                # The chip runs FOUR hardware queues side by side.  Whether the per-update all-gather keeps a fifth busy depends on the
                # torch build (>= 2.8 launches a synchronous collective on the current = the slot's stream; older ones on the
                # communicator's own), and no multi-GPU box was available to the builder: so the multi-GPU line finds out, untimed,
                # before the warm-up — 16 updates with four slots, 16 with three, every rank timing its own, the job's figure being the
                # max over ranks like `value` — and keeps three slots only if they are clearly (> 3 %) faster.
                def calibrate(n_slots):
                    pipe.set_active(n_slots)
                    for i in range(2 * n_slots):
                        pipe_step(i)
                    pipe_fence()
                    t_c = torch.tensor([timed_updates(None, pipe_step, pipe_fence, 2, 16, dev)], dtype=torch.float64, device=dev)
                    dist.all_reduce(t_c, op=dist.ReduceOp.MAX)
                    return float(t_c.item())
                t_four, t_three = calibrate(4), calibrate(3)
                slots_calibration = dict(ms_16_updates_four_slots=round(t_four * 1e3, 3), ms_16_updates_three_slots=round(t_three * 1e3, 3))
                pipe.set_active(3 if t_three < 0.97 * t_four else 4)
                slots_calibration["chosen"] = pipe.active
            elapsed = timed_updates(None, pipe_step, pipe_fence, args.warmup, args.steps, dev)
            last = (pipe.submitted - 1) % pipe.active
            v.copy_(pipe.v[last]); status.copy_(pipe.status[last])
            if do_gather:
                v_all.copy_(v_all_slots[last])
        else:
            elapsed = timed_updates(eng, step, fence, args.warmup, args.steps, dev)
        rank_ms = None
        if multi:
            rank_ms = rank_spread(elapsed)              # every rank's own clock: a straggler shows as max >> min
            elapsed = rank_ms["max"] * args.steps * 1e-3
        status_host = status.cpu().numpy().copy()
        v_host = v_slots[(args.warmup + args.steps - 1) & 1].cpu().numpy().copy() if async_gather else v.cpu().numpy().copy()
        gathered_ok, ranks_seen = None, None
        if do_gather and not async_gather:
            ok = torch.tensor([int(torch.equal(v_all[rank * B:(rank + 1) * B], v))], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # EVERY rank finds its own rows in the gathered table
            gathered_ok = bool(ok.item())
        if multi:
            # ranks the communicator really spans: every rank contributes a one, the sum must be the world size
            seen = torch.ones(1, dtype=torch.int32, device=dev)
            dist.all_reduce(seen, op=dist.ReduceOp.SUM)
            ranks_seen = int(seen.item())

        # what one update takes on its queue while the others share the chip (device time between two events on the slot's stream)
        lat_in_flight = None
        if pipe is not None and not multi:
            evs = []
            for i in range(10 * in_flight):
                st_k = pipe.streams[pipe.submitted % in_flight]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st_k)
                pipe_step(i)
                e1.record(st_k)
                evs.append((e0, e1))
            pipe_fence()
            lat_in_flight = float(np.median([a.elapsed_time(b) for a, b in evs[2 * in_flight:]]))

        # single-update latency with a host synchronisation per update (a control loop's view)
        lat = []
        for i in range(10):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            step(args.warmup + i % max(args.steps, 1))
            torch.cuda.synchronize(dev)
            lat.append(time.perf_counter() - t1)

        # instrumented pass: HIP event pairs around every launch, on the launch stream, one update at a time (kernels of
        # several queues share the chip: a kernel's own duration is only defined while it runs alone).  With several updates
        # in flight it runs on one of the pipeline's handles, i.e. the kernels and the tile plan `value` was measured with.
        eng_prof = pipe.engines[0] if pipe is not None else eng
        prof_step = make_step(eng_prof)
        eng.lib.vitvs_op_plan_in_flight(in_flight)      # the operator hooks below (tile symbol, plain chain) plan alike
        eng_prof.timing_enable(True)
        n_prof = min(args.steps, 50)
        for i in range(n_prof):
            prof_step(args.warmup + i)
        if gather is not None:
            gather.finish()
        prof = eng_prof.timing_collect()
        eng_prof.timing_enable(False)
        plain = None
        if rank == 0 and world == 1 and not args.no_plain_chain:   # single-process extra; ranks stay in lock step at N > 1
            m_rows, bk_ = 2 * B * cfg.seq, 128 // ELEM_BYTES[args.precision]
            plain = {name: plain_chain_us(args.precision, m_rows, cfg.dim, kk, split_k(m_rows, cfg.dim, kk, bk_, in_flight, args.precision != "fp32"), dev)
                     for name, kk in (("proj", cfg.dim), ("fc2", cfg.hidden))}

        overlapped = None
        if rank == 0 and world == 1 and not args.no_plain_chain and in_flight > 1:
            m_rows, bk_ = 2 * B * cfg.seq, 128 // ELEM_BYTES[args.precision]
            try:                                     # an auxiliary figure: it must never cost the line
                overlapped = overlapped_chain_us(args.precision, m_rows, cfg.dim, cfg.hidden, split_k(m_rows, cfg.dim, cfg.dim, bk_, in_flight, args.precision != "fp32"),
                                                 split_k(m_rows, cfg.dim, cfg.hidden, bk_, in_flight, args.precision != "fp32"), dev, queues=in_flight)
            except Exception as exc:                 # noqa: BLE001
                print(f"bench.py: roofline.overlapped not measured ({type(exc).__name__}: {exc})", file=sys.stderr)
                torch.cuda.synchronize(dev)
        eng.lib.vitvs_op_plan_in_flight(1)              # the thread's hint back to its default: handles created below plan for themselves

        parity = None
        secondary = None
        if world == 1 and rank == 0 and not args.no_cpu_baseline and not dense:
            # (the handle and tile plan `value` was measured with: a pipeline slot when several updates are in flight)
            parity = parity_block(eng_prof, cfg, sd, params, des_np[0], cur_np[0], depth_np, I_cur, I_des, Z, K, orders[0, 0], _lib)
        if world == 1 and rank == 0 and not args.no_secondary and args.precision in ("bf16", "fp16") and not dense:
            # The parity mode at servo rate: split-f16 (VITVS_F16X2: every operand an fp16 hi / lo pair, three f16 MFMAs per
            # k-step) — the precision every fp32-mode parity test also runs under, unchanged bars (tests/test_gpu_path.py EXACT) —
            # same process, same inputs and orders, one update in flight and `in_flight` in flight like `value`; the fp32
            # matrix pipe's own rate (the mode of the earlier rounds' `secondary`) beside it.
            eng_x2 = Engine(cfg, params, precision="f16x2", max_pairs=B).load_state_dict(sd)
            s_steps, s_warm = max(10, args.steps // 2), max(3, args.warmup // 2)
            el_x2 = timed_updates(eng_x2, make_step(eng_x2), fence, s_warm, s_steps, dev)
            secondary = dict(dtype="f16x2", metric="servo_updates_per_sec", value_one_in_flight=round(B * s_steps / el_x2, 2),
                             unit="updates/s", steps=s_steps, warmup=s_warm, ms_per_step_one_in_flight=round(el_x2 / s_steps * 1e3, 4),
                             note="split-f16 parity mode: fp32-class results (bit-exact arg-max tables on the strict fixtures, tokens "
                                  "<= 2e-5, v_c <= 1e-9: the fp32 tests of tests/test_gpu_path.py run under it unchanged) on the f16 "
                                  "matrix cores; operand bytes as fp32, three MFMAs per k-step")
            if not args.no_cpu_baseline:
                secondary["parity"] = parity_block(eng_x2, cfg, sd, params, des_np[0], cur_np[0], depth_np, I_cur, I_des, Z, K,
                                                   orders[0, 0], _lib)
            eng_x2.close()
            if pipe is not None:
                # (on the main pipeline's streams, idle now: three more high-priority streams would share the class's four hardware
                # queues with them, and two slots that land on one queue take turns — measured 1614 instead of 2367 updates/s)
                pipe_x2 = UpdatePipeline(cfg, params, sd, precision="f16x2", depth=in_flight, max_pairs=B, device=dev, streams=pipe.streams)

                def x2_step(i):
                    pipe_x2.submit(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % total], None, False, inputs_ready=True)

                def x2_fence():
                    pipe_x2.synchronize()
                    torch.cuda.synchronize(dev)
                for i in range(2 * in_flight):
                    x2_step(i)
                x2_fence()
                el_p = timed_updates(None, x2_step, x2_fence, s_warm, s_steps, dev)
                secondary.update(value=round(B * s_steps / el_p, 2), ms_per_step=round(el_p / s_steps * 1e3, 4),
                                 updates_in_flight=in_flight)
                pipe_x2.close()
            else:
                secondary.update(value=secondary["value_one_in_flight"], ms_per_step=secondary["ms_per_step_one_in_flight"],
                                 updates_in_flight=1)
            eng32 = Engine(cfg, params, precision="fp32", max_pairs=B).load_state_dict(sd)
            f_steps32, f_warm32 = max(10, args.steps // 8), 3
            el32 = timed_updates(eng32, make_step(eng32), fence, f_warm32, f_steps32, dev)
            secondary["fp32_matrix_pipe"] = dict(value_one_in_flight=round(B * f_steps32 / el32, 2), steps=f_steps32,
                                                 ms_per_step=round(el32 / f_steps32 * 1e3, 4),
                                                 note="the same updates on v_mfma_f32_16x16x4_f32 (157 TFLOP/s dense peak -> a ceiling "
                                                      "of ~2.2 k updates/s): what the split-f16 mode replaces as the parity mode")
            eng32.close()

        # NOT the headline: the same updates with the goal frame's tokens computed once (vitvs_set_goal_dev) — what a servo
        # loop with a fixed goal image can use; the reference (and `value`) recompute them on every update
        goal_cached = None
        if world == 1 and rank == 0 and not args.no_secondary and not dense:
            eng.set_goal(I_des)
            c_steps, c_warm = max(10, args.steps // 2), max(3, args.warmup // 2)
            elc = timed_updates(eng, make_step(eng, "cached"), fence, c_warm, c_steps, dev)
            goal_cached = dict(metric="servo_updates_per_sec", value=round(B * c_steps / elc, 2), unit="updates/s", steps=c_steps,
                               warmup=c_warm, ms_per_step=round(elc / c_steps * 1e3, 4), dtype=args.precision,
                               note="one update in flight, goal tokens cached (vitvs_set_goal_dev): only the current frame is forwarded; the "
                                    "reference recomputes I_des every update, and so does `value`")

        # The boundary's host-buffer form (vitvs_compute_velocity: frames, depth, intrinsics and the visiting order in HOST memory,
        # v_c back in host memory, one synchronous call per update): the PCIe-inclusive rate of one control loop.  Never `value`.
        host_buffers = None
        if world == 1 and rank == 0 and not args.no_secondary and not dense:
            import ctypes as C
            hp = lambda a: a.ctypes.data_as(C.c_void_p)                       # noqa: E731
            k_host = np.ascontiguousarray(np.array([params.intrinsics()] * B, np.float64))
            z_host = np.ascontiguousarray(np.stack([depth_np] * B))
            orders_host = orders[:8].cpu().numpy().copy()
            v_out, st_out = np.zeros((B, 6), np.float64), np.zeros(B, np.int32)
            eng.set_option("in_flight", 1)

            # (argument pointers built once: a C caller passes addresses, and eight numpy -> ctypes conversions per call are ~10 us of
            #  interpreter time inside a 0.5 ms call)
            p_cur, p_des, p_z, p_k, p_v, p_st = hp(cur_np), hp(des_np), hp(z_host), hp(k_host), hp(v_out), hp(st_out)
            p_orders = [hp(orders_host[j]) for j in range(8)]
            call_host = eng.lib.vitvs_compute_velocity

            def host_step(i):
                rc = call_host(eng.handle, B, p_cur, p_des, 0, p_z, p_k, _lib.SELECT_ORDER, p_orders[i % 8], None, 0, p_v, p_st)
                if rc != 0:
                    raise RuntimeError(f"vitvs_compute_velocity failed ({rc})")
            h_steps = max(10, args.steps // 2)
            for i in range(3):
                host_step(i)
            t_h = time.perf_counter()
            for i in range(h_steps):
                host_step(3 + i)
            el_h = time.perf_counter() - t_h
            host_buffers = dict(metric="servo_updates_per_sec", value=round(B * h_steps / el_h, 2), unit="updates/s", steps=h_steps,
                                ms_per_step=round(el_h / h_steps * 1e3, 4), dtype=args.precision,
                                bytes_over_pcie_per_update=int(B * (2 * cfg.img_size ** 2 * 3 + depth_np.nbytes + 32 + cfg.tokens * 4 + 52)),
                                note="host-pointer entry point (vitvs_compute_velocity), pageable numpy buffers in, v_c out, one synchronous "
                                     "call per update on one handle: memcpy into the handle's pinned block, one copy launch for the frames, "
                                     "the depth image copied behind the forward's launches and read in place by the law (the PCIe-inclusive "
                                     "form of `sequential`); `value` is device-resident by contract")

        # The drop-in's own rate: the reference's Controller.ibvs() (vitvs_v2.py:588-632) through vit-vs_amd/servo.py — camera
        # frames (u_max x v_max uint8, the fused Pillow resize) and the uint16 depth image as HOST numpy arrays, handed over by the
        # callbacks before every step like the reference's ROS callbacks do; one handle, one update at a time, EMA and feature
        # arrays included.  selection "order": the device-side draw (one C call per update); "reference": the reference's own
        # sort + randperm draw on the host between the correspondence and the law (two device round trips).
        controller_loop = None
        if world == 1 and rank == 0 and not args.no_secondary and not dense and B == 1:
            from PIL import Image
            from vitvs_amd import servo as vservo
            cam = lambda a: np.asarray(Image.fromarray(a).resize((params.u_max, params.v_max)), dtype=np.uint8)   # noqa: E731
            goal_cam, cur_cam = cam(des_np[0]), cam(cur_np[0])
            eng_c = Engine(cfg, params, precision=args.precision, max_pairs=1).load_state_dict(sd)
            controller_loop = dict(frames=f"{params.u_max}x{params.v_max} uint8 host arrays + uint16 depth, fused Pillow-exact resize",
                                   dtype=args.precision, note="Controller.ibvs() per update incl. the callbacks' hand-over, the draw, "
                                   "EMA and the feature arrays the reference's detect_features returns; one update in flight")
            eng_cx = Engine(cfg, params, precision="f16x2", max_pairs=1).load_state_dict(sd) if args.precision != "f16x2" else None
            for sel_name, eng_l in (("order", eng_c), ("reference", eng_c), ("order_f16x2", eng_cx), ("reference_f16x2", eng_cx)):
                if eng_l is None:
                    continue
                ctl = vservo.Controller(eng_l, goal_image=goal_cam, selection=sel_name.split("_")[0])
                ctl.generator = torch.Generator().manual_seed(121)
                n_loop, lat_c = 230, []
                for i in range(n_loop):
                    ctl.image_callback_rgb(cur_cam)
                    ctl.image_callback_depth(depth_np)
                    t_c = time.perf_counter()
                    ctl.ibvs()
                    lat_c.append(time.perf_counter() - t_c)
                lat_c = np.array(lat_c[30:]) * 1e3
                controller_loop[sel_name] = dict(updates=len(lat_c), updates_per_s=round(1e3 / float(lat_c.mean()), 1),
                                                 median_ms=round(float(np.median(lat_c)), 4), p90_ms=round(float(np.percentile(lat_c, 90)), 4),
                                                 v_c_is_set=ctl.v_c is not None, status=ctl.last_status)
            eng_c.close()
            if eng_cx is not None:
                eng_cx.close()

        # fp16: the same kernels on v_mfma_f32_16x16x32_f16 — the throughput dtype for trained checkpoints (DESIGN.md section 3:
        # 96-99.5 % arg-max agreement on trained-like weights where bf16 keeps 77-91 %); same protocol as `value`, fewer steps
        same_kernels_fp16 = None
        if world == 1 and rank == 0 and not args.no_secondary and not dense and args.precision == "bf16" and pipe is not None:
            pipe16 = UpdatePipeline(cfg, params, sd, precision="fp16", depth=in_flight, max_pairs=B, device=dev, streams=pipe.streams)

            def step16(i):
                pipe16.submit(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % total], None, False, inputs_ready=True)

            def fence16():
                pipe16.synchronize()
                torch.cuda.synchronize(dev)
            for i in range(2 * in_flight):
                step16(i)
            fence16()
            f_steps, f_warm = max(20, args.steps // 2), max(5, args.warmup // 2)
            el16 = timed_updates(None, step16, fence16, f_warm, f_steps, dev)
            same_kernels_fp16 = dict(metric="servo_updates_per_sec", value=round(B * f_steps / el16, 2), unit="updates/s", dtype="fp16",
                                     steps=f_steps, warmup=f_warm, ms_per_step=round(el16 / f_steps * 1e3, 4),
                                     note=f"{in_flight} updates in flight, fp16 operands: the recommended throughput dtype for trained "
                                          "checkpoints (tests/test_gpu_path.py::test_trained_like_statistics_end_to_end)")
            pipe16.close()

    updates = world * B * args.steps
    value = updates / elapsed
    es = ELEM_BYTES[args.precision]
    big_rule = args.precision != "fp32"
    work = kernel_work(cfg, 2 * B, B, es, binned, in_flight, big_rule)
    kernels = {}
    for name, (ms, cnt) in prof.items():
        if cnt == 0:
            continue
        avg = max(ms / cnt * 1e-3, 1e-7)
        fl, by = work[name]
        kernels[name] = dict(launches_per_step=cnt / n_prof, avg_us=round(avg * 1e6, 3),
                             step_share_us=round(avg * cnt / n_prof * 1e6, 2),
                             tflops=round(fl / avg / 1e12, 3), gbps=round(by / avg / 1e9, 1))
    # roofline object: the kernel SYMBOL with the largest share of the step (proj and fc2 are the same
    # split-K kernel, as rocprofv3 --stats reports them), priced with its algorithmic work per launch
    prec_tag = {"bf16": "bf16", "fp16": "f16", "fp32": "f32", "f16x2": "hx2"}[args.precision]
    groups = {"linear_partial(proj+fc2)": ["proj", "fc2"]}
    for k in kernels:
        if k not in ("proj", "fc2"):
            groups[k] = [k]

    def g_share(g):
        return sum(kernels[c]["step_share_us"] for c in groups[g] if c in kernels)
    dom = max(groups, key=g_share)
    members = [c for c in groups[dom] if c in kernels]
    launches = sum(kernels[c]["launches_per_step"] for c in members)
    avg_s = g_share(dom) / launches * 1e-6
    fl = sum(work[c][0] * kernels[c]["launches_per_step"] for c in members) / launches
    by = sum(work[c][1] * kernels[c]["launches_per_step"] for c in members) / launches

    def linear_symbol(m, n, kk, s_, partial):   # the kernel symbol of the tile the LIBRARY reports for this layer (no mirror)
        import ctypes
        tile = (ctypes.c_int32 * 3)()
        prec_id = {"f32": _lib.F32, "bf16": _lib.BF16, "f16": _lib.F16, "hx2": _lib.F16X2}[prec_tag]
        prev_hint = eng.lib.vitvs_op_plan_in_flight(in_flight)       # the plan `value` ran under; restored at once
        rc_ = eng.lib.vitvs_op_linear_tile(prec_id, m, n, kk, s_ if partial else 0, tile)
        eng.lib.vitvs_op_plan_in_flight(prev_hint)
        if rc_ != 0:
            return None
        if tile[2] == 0:
            return f"linear_big_kernel<{prec_tag},{tile[0]}x{tile[1]}>:" + ("BigPartial" if partial else "BigStore")
        return f"linear_kernel<{prec_tag},{tile[0]},{tile[1]},{tile[2]}>:" + ("EpiPartial" if partial else "EpiStore")
    def attention_symbol(tag, cfg_, b_):     # mirror of launch_attention's dispatch (attention.hip)
        short = cfg_.seq <= 256 and -(-cfg_.seq // 16) * cfg_.heads * 2 * b_ <= 640
        if tag == "f32":
            return "attention_f32_kernel"
        if tag == "hx2":
            return "attention_x2_short_kernel" if short else ("attention_x2_long_kernel" if cfg_.seq >= 2048 else "attention_x2_kernel")
        if short:
            return f"attention_16_short_kernel<{tag}>"
        if cfg_.seq >= 512 or (cfg_.seq >= 128 and -(-cfg_.seq // 64) * cfg_.heads * 2 * b_ > 256):
            return f"attention_16_long_kernel<{tag}>"
        return f"attention_16_kernel<{tag}>"
    m_all = 2 * B * cfg.seq
    bk_es = 128 // es
    symbol = {"linear_partial(proj+fc2)": linear_symbol(m_all, cfg.dim, cfg.hidden, split_k(m_all, cfg.dim, cfg.hidden, bk_es, in_flight, big_rule), True),
              "residual_ln": f"residual_ln_kernel<{prec_tag}>",
              "fc1": linear_symbol(m_all, cfg.hidden, cfg.dim, 1, False),
              "qkv": linear_symbol(m_all, 3 * cfg.dim, cfg.dim, 1, False),
              "attention": attention_symbol(prec_tag, cfg, B)}.get(dom, dom)
    # HBM traffic (PMC) comes from separate rocprofv3 --pmc passes of this same command, never from this run: the line
    # says which committed file and which commit of the kernels it was measured on, and mixes it into no live ratio.
    traffic, traffic_source = None, None
    TRAFFIC_PROFILE = _profile(TRAFFIC_PROFILES.get(args.precision, "none"))
    STATS_PROFILE = _profile(STATS_PROFILES.get(args.precision, "none"))
    pmc_path = os.path.join(ROOT, TRAFFIC_PROFILE)
    if os.path.isfile(pmc_path) and args.precision in TRAFFIC_PROFILES and args.config == "vitb16_224" and B == 1:
        with open(pmc_path) as fh:
            blob = json.load(fh)
        entry = blob.get("kernels", blob).get(symbol)
        if entry:
            traffic = entry.get("hbm_bytes_per_launch")
            traffic_source = dict(file=TRAFFIC_PROFILE, symbol=symbol, measured_at_commit=blob.get("git_commit"),
                                  command=blob.get("command"), mfma_busy_cycles_per_launch=entry.get("mfma_busy_cycles_per_launch"),
                                  note="separate --pmc FETCH_SIZE / WRITE_SIZE (and SQ_VALU_MFMA_BUSY_CYCLES) passes; "
                                       "FETCH_SIZE doubled per the gfx950 correction")
    # the committed rocprofv3 --stats summary of the same command: average duration of the same symbol, so that the line and
    # profiles/ can be reconciled mechanically (it is a record of ANOTHER run, on another box of the pool: +-4 %)
    from_profile = None
    stats_path = os.path.join(ROOT, STATS_PROFILE)
    if os.path.isfile(stats_path) and args.precision in STATS_PROFILES and args.config == "vitb16_224" and B == 1 and symbol:
        import csv
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from symbols import short as short_symbol          # rocprofv3's (mangled) kernel names -> the names used here
        with open(stats_path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Name", "")
                if short_symbol(name) == symbol:
                    calls, avg_ns = int(row["Calls"]), float(row["AverageNs"])
                    from_profile = dict(file=STATS_PROFILE, name=name[:120], calls=calls, avg_ns=round(avg_ns, 1),
                                        total_ms=round(calls * avg_ns * 1e-6, 3))
                    break
    mfma_bound = dom in ("qkv", "fc1", "attention", "patch_embed", "gram_argmax", "linear_partial(proj+fc2)")
    if mfma_bound:
        peak = PEAK_MFMA["fp32" if dom == "gram_argmax" else args.precision]
        roof = dict(kernel=symbol, classes=members, bound="mfma", achieved=round(fl / avg_s / 1e12, 3), peak=peak / 1e12,
                    unit="TFLOP/s", frac=round(fl / avg_s / peak, 5), traffic=traffic, traffic_source=traffic_source,
                    algorithmic_flops_per_launch=fl, algorithmic_bytes_per_launch=by,
                    avg_launch_us=round(avg_s * 1e6, 3))
        if args.precision == "f16x2" and dom != "gram_argmax":
            roof["executed_flops_per_launch"] = 3 * fl
            roof["frac_of_split_ceiling"] = round(3 * fl / avg_s / peak, 5)
            roof["note"] = ("split-f16: hi.hi + hi.lo + lo.hi = three f16 MFMAs per algorithmic MAC, so peak / 3 is the scheme's "
                            "ceiling; `achieved` / `frac` price the ALGORITHMIC FLOPs against the full f16 peak")
        if from_profile:
            from_profile["achieved"] = round(fl / (from_profile["avg_ns"] * 1e-9) / 1e12, 3)
            from_profile["frac"] = round(fl / (from_profile["avg_ns"] * 1e-9) / peak, 5)
            roof["recomputed_from_profile"] = from_profile
        if plain and dom == "linear_partial(proj+fc2)":
            # the same kernel in a chain of plain launches (what the un-instrumented step pays per launch)
            p_us = sum(plain[c] * kernels[c]["launches_per_step"] for c in members) / launches
            roof.update(plain_launch_us=round(p_us, 3), achieved_plain=round(fl / (p_us * 1e-6) / 1e12, 3),
                        frac_plain=round(fl / (p_us * 1e-6) / peak, 5))
        if overlapped and dom == "linear_partial(proj+fc2)":
            # ... and with `in_flight` queues busy, the regime `value` is measured in: proj and fc2 alternate 1 : 1 there, as in
            # the forward (the patch-embedding launch, 1 of the symbol's 25 per update, is left out), so the FLOPs per launch
            # are their mean
            fl_o = (work["proj"][0] + work["fc2"][0]) / 2
            o_us = overlapped["per_launch_us"]
            overlapped.update(achieved=round(fl_o / (o_us * 1e-6) / 1e12, 3), unit="TFLOP/s", frac=round(fl_o / (o_us * 1e-6) / peak, 5),
                              achieved_one_queue=round(fl_o / (overlapped["per_launch_us_one_queue"] * 1e-6) / 1e12, 3),
                              algorithmic_flops_per_launch=fl_o,
                              note="the dominant symbol as a captured chain of proj / fc2 launches in forward order on each of the "
                                   "queues (random operands, 12 shared weight sets): wall time / launches over all queues")
            roof["overlapped"] = overlapped
    else:
        roof = dict(kernel=symbol, classes=members, bound="hbm", achieved=round(by / avg_s / 1e9, 2), peak=PEAK_HBM / 1e9,
                    unit="GB/s", frac=round(by / avg_s / PEAK_HBM, 5), traffic=traffic, traffic_source=traffic_source,
                    algorithmic_bytes_per_launch=by, avg_launch_us=round(avg_s * 1e6, 3))
        if from_profile:
            from_profile["achieved"] = round(by / (from_profile["avg_ns"] * 1e-9) / 1e9, 2)
            from_profile["frac"] = round(by / (from_profile["avg_ns"] * 1e-9) / PEAK_HBM, 5)
            roof["recomputed_from_profile"] = from_profile

    if host_buffers:
        # a host-pointer call ends with the caller holding v_c, i.e. with a wait: its device-resident counterpart is the
        # synchronised single update (latency_ms_with_host_sync), not the back-to-back stream of `sequential`
        sync_ms = float(np.median(lat)) * 1e3
        host_buffers["vs_sequential"] = round(host_buffers["value"] / (sequential["value"] if sequential else value), 4)
        host_buffers["vs_device_resident_update_with_host_sync"] = round(sync_ms / host_buffers["ms_per_step"], 4)
    out = dict(
        metric="servo_updates_per_sec", value=round(value, 2), unit="updates/s", n_gpus=world, steps=args.steps,
        warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4), higher_is_better=True, scaling="weak",
        vs_baseline=None, dtype=args.precision, data="synthetic",
        config=dict(workload=f"{cfg.model_type} {cfg.img_size}x{cfg.img_size} frame pair(s): both frames forwarded "
                             f"through block {cfg.layer}, cosine correspondence, mutual-NN, {params.num_pairs} features "
                             f"in a fresh random order, L_e, pinv -> v_c; I_des recomputed every update; "
                             f"{'3x3 log-binned descriptors (9 D wide); ' if binned else ''}"
                             f"{B} pair(s) per update, {pipe.active if pipe is not None else 1} independent update(s) in flight",
                    key=args.config, binned=binned, pairs_per_step_per_gpu=B, updates_in_flight_per_gpu=(pipe.active if pipe is not None else 1), tokens=cfg.tokens, dim=cfg.dim,
                    parallelism=(f"dp{world} (frame pairs sharded, " + ("no collective" if not do_gather else "v_c all-gather per step")
                                 + f"{', asynchronous' if async_gather else ''})") if world > 1 else "single GPU",
                    weights="synthetic seed 0", frame_seeds=seeds, selection="DENSE" if dense else "ORDER"),
        protocol=(f"{pipe.active} independent batch-{B} updates in flight per GPU (one handle + one high-priority stream each, shared "
                  f"weights, hipGraph replay, the in_flight tile plan); the timed region is K updates between two barrier + "
                  f"synchronize pairs, filling and draining the {pipe.active} slots included" if in_flight > 1 else
                  "one update in flight per GPU: one handle, one stream, plain launches"),
        value_one_in_flight=(sequential["value"] if sequential else round(value, 2)),
        roofline=roof,
        cpu_baseline=None,
        parity=parity,
        sequential=sequential,
        secondary=secondary,
        goal_cached=goal_cached,
        host_buffers=host_buffers,
        controller_loop=controller_loop,
        same_kernels_fp16=same_kernels_fp16,
        path=dict(gflop_per_update=round(cfg.flops_per_pair(binned) / 1e9, 3),
                  tflops=round(cfg.flops_per_pair(binned) * value / world / 1e12, 3),
                  frac_of_mfma_peak=round(cfg.flops_per_pair(binned) * value / world / PEAK_MFMA[args.precision], 5),
                  weight_bytes=cfg.weight_elems() * es,
                  weight_stream_frac_of_hbm_peak=round(cfg.weight_elems() * es * value / world / B / PEAK_HBM, 5)),
        latency_ms_with_host_sync=round(float(np.median(lat)) * 1e3, 4),
        latency_ms_per_update_in_flight=None if lat_in_flight is None else round(lat_in_flight, 4),
        kernels=kernels,
        status=[int(s) for s in status_host],
        v_c=[float(x) for x in v_host[0]],
    )
    if rank_ms is not None:
        out["per_rank_ms_per_step"] = dict(rank_ms, max=round(rank_ms["max"], 4))
        out["v_c_gather"] = ("none (--no-gather: every rank keeps its own twists)" if not do_gather else
                             "one all-gather of 6 doubles per pair behind every update")
    if slots_calibration is not None:
        out["slots_calibration"] = dict(slots_calibration, note="untimed, before the warm-up: 16 updates (all-gather included) with four "
                                        "slots and with three, max over ranks; three are kept only if > 3 % faster — the chip runs four "
                                        "hardware queues side by side and a collective on a communicator stream of its own would be a fifth")
    if gathered_ok is not None:
        out["gathered_rows_match_local"] = gathered_ok
    if ranks_seen is not None:
        out["rccl_ranks_seen"] = ranks_seen
        out["dist_backend"] = os.environ.get("VITVS_DIST_BACKEND", "nccl")
        if ranks_seen != world:
            raise SystemExit(f"the communicator spans {ranks_seen} ranks, expected {world}")
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, sd, des_np[0], cur_np[0], depth_np, params)
    if rank == 0:
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
