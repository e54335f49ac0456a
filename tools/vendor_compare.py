"""Vendor-library anchor for the roofline claims (tools only, never product): what hipBLASLt (through ``torch.matmul`` /
``F.linear``) and PyTorch's fused attention (``F.scaled_dot_product_attention``) do on the SAME shapes, the same box and the
same random bf16 data as ``tools/big_ops``, next to this library's own kernels called through the C ABI.

    python tools/vendor_compare.py [--fast] > profiles/r03_vendor.txt

Each line: shape | vendor us / TFLOP/s | ours us / TFLOP/s | ours / vendor.  The vendor GEMM computes A W^T + bias (no GELU,
bf16 out): for the fc1 layers ours also applies GELU in its epilogue, i.e. does strictly more in the time shown.  Timing:
`reps` back-to-back launches captured into one hipGraph and replayed (neither side pays host launch cost), 4 rotating
weight sets (so that W is not L2-resident from the previous launch, as in the forward), best of 5 replays.
"""
import ctypes as C
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import vitvs_amd  # noqa: E402,F401
from vitvs_amd import _lib  # noqa: E402


def timed(fn, reps):
    """us per call: `reps` calls captured into one hipGraph and replayed (no host launch cost on either side: torch's
    Python dispatch is ~15 us per call, more than most of these kernels take), best of 5 replays; eager timing if the
    capture fails."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):          # warm up ON the capture stream (the operator hook's workspace is per stream)
        for _ in range(5):
            fn()
    torch.cuda.synchronize()
    graph = None
    try:
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(reps):
                    fn()
        graph = g
    except Exception as exc:  # noqa: BLE001
        print(f"# graph capture failed ({type(exc).__name__}: {exc}); eager timing", file=sys.stderr)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(reps):
                fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    fast = "--fast" in sys.argv
    reps = 20 if fast else 60
    lib = _lib.load()
    dev = torch.device("cuda")
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
    p = lambda t: C.c_void_p(t.data_ptr())                             # noqa: E731
    g = torch.Generator(device="cpu").manual_seed(1)
    print(f"# {torch.cuda.get_device_name(0)}, torch {torch.__version__}, bf16, random operands (uniform), {reps} launches per graph replay, best of 5")
    print(f"{'GEMM  M x N x K':34s} | {'vendor (torch F.linear)':>24s} | {'ours (C ABI, auto tile)':>24s} | ours/vendor time")
    shapes = [("ViT-B/8 448  qkv", 6274, 2304, 768, 0, 0), ("ViT-B/8 448  fc1 (+GELU ours)", 6274, 3072, 768, 1, 0),
              ("ViT-B/8 448  fc2", 6274, 768, 3072, 0, -1), ("ViT-B/8 448  proj", 6274, 768, 768, 0, -1),
              ("8 x ViT-B/16 qkv", 3152, 2304, 768, 0, 0), ("8 x ViT-B/16 fc1 (+GELU ours)", 3152, 3072, 768, 1, 0),
              ("8 x ViT-B/16 fc2", 3152, 768, 3072, 0, -1), ("8 x ViT-B/16 proj", 3152, 768, 768, 0, -1),
              ("ViT-L/14 518 qkv", 2740, 3072, 1024, 0, 0), ("ViT-L/14 518 fc1 (+GELU ours)", 2740, 4096, 1024, 1, 0),
              ("ViT-L/14 518 fc2", 2740, 1024, 4096, 0, -1), ("ViT-L/14 518 proj", 2740, 1024, 1024, 0, -1),
              ("1 pair ViT-B/16 qkv", 394, 2304, 768, 0, 0), ("1 pair ViT-B/16 fc1 (+GELU ours)", 394, 3072, 768, 1, 0),
              ("1 pair ViT-B/16 fc2", 394, 768, 3072, 0, -1), ("1 pair ViT-B/16 proj", 394, 768, 768, 0, -1)]
    for name, M, N, K, gelu, slices in shapes:
        A = (torch.rand((M, K), generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
        Ws = [((torch.rand((N, K), generator=g) * 2 - 1) * 0.05).to(torch.bfloat16).to(dev) for _ in range(4)]
        bias = torch.zeros(N, dtype=torch.float32, device=dev)
        bias16 = bias.to(torch.bfloat16)
        turn = [0]

        def vendor():
            turn[0] += 1
            return F.linear(A, Ws[turn[0] % 4], bias16)
        sl = lib.vitvs_op_splitk_slices(_lib.BF16, M, N, K) if slices < 0 else 0
        out = torch.empty((max(sl, 1) * M * N * (2 if sl else 1),), dtype=torch.bfloat16, device=dev)   # fp32 slices when split

        def ours():
            turn[0] += 1
            rc = lib.vitvs_op_linear_variant(_lib.BF16, 0, p(A), p(Ws[turn[0] % 4]), p(bias), p(out), M, N, K, gelu, sl, st())
            assert rc == 0
        tv, to = timed(vendor, reps), timed(ours, reps)
        fl = 2.0 * M * N * K
        note = f" ({sl} K slice(s), fp32 partial sums out)" if sl else ""
        print(f"{name:34s} {M:5d} x {N:4d} x {K:4d} | {tv:8.1f} us {fl / tv * 1e-6:7.0f} TF | {to:8.1f} us {fl / to * 1e-6:7.0f} TF | {to / tv:5.2f}{note}")
    print()
    print(f"{'attention  images x tokens x heads (head dim 64)':50s} | {'vendor (torch SDPA)':>22s} | {'ours (C ABI)':>22s} | ours/vendor time")
    for n_img, N, H in [(2, 3137, 12), (4, 3137, 12), (2, 1370, 16), (2, 785, 12), (8, 785, 12), (16, 197, 12), (2, 197, 12)]:
        D = H * 64
        qkv = (torch.rand((n_img * N, 3 * D), generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
        q, k, v = (t.transpose(1, 2) for t in qkv.view(n_img, N, 3, H, 64).unbind(2))     # [B, H, N, 64] strided views
        out = torch.empty((n_img * N, D), dtype=torch.bfloat16, device=dev)

        def vendor():
            return F.scaled_dot_product_attention(q, k, v)

        def ours():
            assert lib.vitvs_op_attention(_lib.BF16, p(qkv), p(out), n_img, N, H, st()) == 0
        try:
            tv = timed(vendor, max(reps // 2, 5))
        except Exception as exc:  # noqa: BLE001
            tv = float("nan")
            print(f"# SDPA failed on {n_img} x {N} x {H}: {exc}")
        to = timed(ours, max(reps // 2, 5))
        fl = 4.0 * n_img * N * N * D
        print(f"{n_img:2d} x {N:4d} x {H:2d} {'':34s} | {tv:8.1f} us {fl / tv * 1e-6:7.0f} TF | {to:8.1f} us {fl / to * 1e-6:7.0f} TF | {to / tv:5.2f}")
        ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(n_img * N, D)
        err = float((out.float() - ref.float()).abs().max())
        assert err < 0.05, err


if __name__ == "__main__":
    main()
