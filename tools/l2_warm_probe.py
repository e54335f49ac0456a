"""Do warm private L2s shorten the one-pair GEMM launches?  ("an isolated load-only kernel
with L2-resident operands runs 2-3x faster".)

Best case for the idea, serially: before each GEMM launch a touch launch makes EVERY XCD's L2 read the GEMM's whole weight
matrix (vitvs_op_touch), the GEMM follows on the same stream, and its duration is taken between two events around it.  The
weights rotate through 12 sets (like the 12 blocks: 12 x 3.5 ... 9.4 MB, far beyond the 8 x 4 MB of L2), so without the touch
they come from the Infinity Cache / HBM as in the forward.  If the touched launch is no shorter here, no concurrent side branch
can make it shorter in the update.

  python tools/l2_warm_probe.py [--precision bf16|f16x2] > profiles/r05_l2_warm_probe.txt
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vitvs_amd  # noqa: E402,F401
from vitvs_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--reps", type=int, default=240)
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    code = {"bf16": _lib.BF16, "fp16": _lib.F16, "f16x2": _lib.F16X2}[args.precision]
    dt = torch.float16 if args.precision != "bf16" else torch.bfloat16
    wide = 2 if args.precision == "f16x2" else 1
    g = torch.Generator().manual_seed(1)
    rnd = lambda r, c: (torch.randn((r, c * wide), generator=g) * 0.05).to(dt).to(dev)   # noqa: E731
    M = 394
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    print(f"precision {args.precision}, M = {M} rows, 12 rotating weight sets, {args.reps} launches per figure; us per launch between "
          f"two events around the GEMM (median | mean); plan hint = updates in flight")
    for hint in (1, 3):
        lib.vitvs_op_plan_in_flight(hint)
        for name, N, K, partial in (("qkv", 2304, 768, False), ("fc1", 3072, 768, False), ("proj", 768, 768, True), ("fc2", 768, 3072, True)):
            ws = [rnd(N, K) for _ in range(12)]
            a = rnd(M, K)
            bias = torch.zeros(N, device=dev)
            slices = lib.vitvs_op_splitk_slices(code, M, N, K) if partial else 0
            out = torch.zeros((max(slices, 1), M, N * (wide if not partial else 1)), dtype=torch.float32 if partial else dt, device=dev)
            wbytes = ws[0].numel() * ws[0].element_size()

            def gemm(w):
                if partial:
                    return lib.vitvs_op_linear_partial(code, P(a), P(w), P(out), M, N, K, slices, st)
                return lib.vitvs_op_linear(code, P(a), P(w), P(bias), P(out), M, N, K, 0, st)
            row = []
            for mode in ("cold", "all XCDs touch all", "each XCD touches its eighth", "touch cost alone"):
                times = []
                for i in range(args.reps + 20):
                    w = ws[i % 12]
                    if mode == "all XCDs touch all":
                        lib.vitvs_op_touch(P(w), wbytes, 0, st)
                    elif mode == "each XCD touches its eighth":
                        lib.vitvs_op_touch(P(w), wbytes, 1, st)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if mode == "touch cost alone":
                        lib.vitvs_op_touch(P(w), wbytes, 0, st)
                    else:
                        assert gemm(w) == 0
                    e1.record()
                    times.append((e0, e1))
                torch.cuda.synchronize()
                t = np.array([x.elapsed_time(y) * 1e3 for x, y in times[20:]])
                row.append(f"{mode}: {np.median(t):6.2f} | {t.mean():6.2f}")
            print(f"  in_flight {hint} {name:4s} {M}x{N}x{K}{' split-K x' + str(slices) if partial else ''} ({wbytes / 1e6:.1f} MB of weights): " + ";  ".join(row))
    lib.vitvs_op_plan_in_flight(1)


if __name__ == "__main__":
    main()
