"""One attention operator launched repeatedly on random operands — the unit rocprofv3 --pmc passes are collected on while working
on the attention kernels (the whole-update bench dilutes one symbol among 86 launches).

  python tools/attn_probe.py [--precision f16x2] [--images 2] [--tokens 3137] [--heads 12] [--reps 10]

Prints the event-timed mean launch duration; under `rocprofv3 --pmc <counters> --output-format csv -- python3 tools/attn_probe.py ...`
the per-dispatch counters of the symbol land in the pass's counter_collection.csv (tools/attn_pmc_table.py puts passes side by side).
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vitvs_amd  # noqa: E402,F401
from vitvs_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16x2", choices=["f16x2", "bf16", "fp16"])
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--tokens", type=int, default=3137)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    n, N, H = a.images, a.tokens, a.heads
    D = 64 * H
    x = torch.randn((n * N, 3 * D), generator=g)
    if a.precision == "f16x2":
        prec = _lib.F16X2
        hi = x.half()
        lo = (x - hi.float()).half()
        qkv = torch.stack([hi.view(n * N, 3 * D // 32, 32), lo.view(n * N, 3 * D // 32, 32)], dim=2).reshape(n * N, 6 * D).contiguous().to(dev)
        out = torch.empty((n * N, 2 * D), dtype=torch.float16, device=dev)
    else:
        prec = _lib.BF16 if a.precision == "bf16" else _lib.F16
        dt = torch.bfloat16 if a.precision == "bf16" else torch.float16
        qkv = x.to(dt).to(dev)
        out = torch.empty((n * N, D), dtype=dt, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.vitvs_op_attention(prec, C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()), n, N, H, st)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    flops = 4.0 * n * N * N * D
    print(f"attention {a.precision} {n} x {N} x {H}: {us:.1f} us per launch (back to back), {flops / us * 1e-6:.0f} TFLOP/s algorithmic")


if __name__ == "__main__":
    main()
