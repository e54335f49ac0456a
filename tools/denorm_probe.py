"""Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL operands on gfx950?  (The split-f16 precision keeps the low halves of
small activations there: DESIGN.md, "split-f16".)  Runs the fp16 GEMM operator on subnormal A / W and prints what came back."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vitvs_amd  # noqa: E402,F401
from vitvs_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    M = N = K = 64
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, a_val, w_val in [("A subnormal 2^-20, W 1024", 2.0 ** -20, 1024.0),
                               ("A 1024, W subnormal 2^-20", 1024.0, 2.0 ** -20),
                               ("A subnormal 2^-24 (smallest), W 4096", 2.0 ** -24, 4096.0),
                               ("both normal 2^-10 x 1", 2.0 ** -10, 1.0)]:
        A = torch.full((M, K), a_val, dtype=torch.float16, device="cuda")
        W = torch.full((N, K), w_val, dtype=torch.float16, device="cuda")
        bias = torch.zeros(N, dtype=torch.float32, device="cuda")
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        rc = lib.vitvs_op_linear(_lib.F16, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(bias.data_ptr()),
                                 C.c_void_p(out.data_ptr()), M, N, K, 0, st)
        torch.cuda.synchronize()
        want = K * a_val * w_val
        print(f"{name}: rc {rc} expected {want:.6g} got {float(out[0, 0]):.6g} (all equal: {bool((out == out[0, 0]).all())})")


if __name__ == "__main__":
    main()
