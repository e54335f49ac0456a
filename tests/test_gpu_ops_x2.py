"""Kernel-level parity (GPU) of the split-f16 precision (VITVS_F16X2, csrc/common.h hx2): the operators of the forward with
every operand an fp16 hi / lo pair and every contraction hi.hi + hi.lo + lo.hi on the f16 matrix cores.

The bar is the fp32 mode's: outputs against an fp64 statement of the op on the SAME fp32 inputs to fp32 rounding
(<= 2e-5 of the output scale; tests/test_gpu_ops.py F32_TOL) — the mode exists to reproduce the reference's fp32
arithmetic (dinov2_extractor.py:245-263) at 16-bit matrix rate, so it gets no 16-bit allowance."""
import ctypes as C
import math

import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import _lib

pytestmark = pytest.mark.gpu

X2 = _lib.F16X2
TOL = 2e-5


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return _lib.load()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _mk(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).float()


def to_x2(t, exp=0):
    """fp32 [R, C] -> fp16 [R, 2C]: per 32 columns [hi | lo] of t * 2^exp (csrc/common.h layout)."""
    r, c = t.shape
    assert c % 32 == 0
    ts = t.float() * (2.0 ** exp)
    hi = ts.clamp(-65504, 65504).half()
    lo = (ts - hi.float()).half()
    return torch.stack([hi.view(r, c // 32, 32), lo.view(r, c // 32, 32)], dim=2).reshape(r, 2 * c).contiguous()


def from_x2(t):
    r, c2 = t.shape
    v = t.view(r, c2 // 64, 2, 32).float()
    return (v[:, :, 0] + v[:, :, 1]).reshape(r, c2 // 2)


def weight_exp(w):
    m = float(w.abs().max())
    return max(0, min(31, 13 - math.frexp(m)[1])) if m > 0 else 0


def test_layout_helpers_round_trip():
    g = torch.Generator().manual_seed(1)
    t = _mk((5, 96), g, 3.0)
    back = from_x2(to_x2(t))
    assert float((back - t).abs().max()) <= 2.0 ** -21 * float(t.abs().max())


@pytest.mark.parametrize("plan", [1, 3])
@pytest.mark.parametrize("M,N,K,gelu", [(394, 2304, 768, 0), (394, 3072, 768, 1), (77, 512, 192, 1), (64, 64, 32, 0), (5, 128, 640, 1),
                                        (197, 1152, 384, 0), (394, 4096, 1024, 1), (1200, 3072, 768, 1), (3170, 2304, 768, 0),
                                        (985, 2304, 768, 0)])
def test_linear(lib, plan, M, N, K, gelu):
    """64 x {64, 96, 128} tiles under both plans (8-wave with two k-groups / 4-wave, staged whole-row epilogue), the 128 x 128
    tiles of the many-row shapes (lane-owned stores), ragged last row tile, K = 32 (one k-tile), weights carrying 2^e."""
    prev = lib.vitvs_op_plan_in_flight(plan)
    try:
        g = torch.Generator().manual_seed(M * 7 + N)
        A = _mk((M, K), g)
        W = _mk((N, K), g, K ** -0.5)
        bias = _mk((N,), g, 0.1)
        ref = A.double() @ W.double().t() + bias.double()
        if gelu:
            ref = torch.nn.functional.gelu(ref)
        e = weight_exp(W)
        Ad, Wd, bd = to_x2(A).cuda(), to_x2(W, e).cuda(), bias.cuda()
        guard = 3
        out = torch.full((M + guard, 2 * N), float("nan"), dtype=torch.float16, device="cuda")
        lib.vitvs_op_weight_exponent(e)
        rc = lib.vitvs_op_linear(X2, _p(Ad), _p(Wd), _p(bd), _p(out), M, N, K, gelu, _stream())
        lib.vitvs_op_weight_exponent(0)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.isnan(out[M:].float()).all(), "rows beyond M were written"
        got = from_x2(out[:M].cpu())
        assert torch.isfinite(got).all()
        assert _rel(got, ref) <= TOL
    finally:
        lib.vitvs_op_plan_in_flight(prev)


def test_linear_small_activations_keep_their_low_halves(lib):
    """Activations of magnitude 2^-6 .. 2^-3: their lo halves are fp16 SUBNORMALS (the activations carry no scale); the matrix
    cores must honour them (tools/denorm_probe.py) or the result drops to 11 bits."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 128, 128, 256
    A = _mk((M, K), g, 2.0 ** -5)
    W = _mk((N, K), g, K ** -0.5)
    ref = A.double() @ W.double().t()
    e = weight_exp(W)
    out = torch.empty((M, 2 * N), dtype=torch.float16, device="cuda")
    bias = torch.zeros(N, device="cuda")
    Ad, Wd = to_x2(A).cuda(), to_x2(W, e).cuda()              # (named: a temporary would be freed before the launch reads it)
    lib.vitvs_op_weight_exponent(e)
    rc = lib.vitvs_op_linear(X2, _p(Ad), _p(Wd), _p(bias), _p(out), M, N, K, 0, _stream())
    lib.vitvs_op_weight_exponent(0)
    assert rc == 0
    torch.cuda.synchronize()
    assert _rel(from_x2(out.cpu()), ref) <= TOL


@pytest.mark.parametrize("plan", [1, 3])
@pytest.mark.parametrize("M,D,K,use_ls,use_ln", [(394, 768, 768, False, True), (394, 768, 3072, True, True), (130, 384, 1536, False, False),
                                                  (197, 1024, 4096, True, True), (1576, 768, 768, True, True), (3152, 768, 3072, False, True),
                                                  (61, 768, 32, False, True)])
def test_split_k_pair(lib, plan, M, D, K, use_ls, use_ln):
    """linear_partial + residual_ln == x + ls * (A W^T + bias), then LayerNorm written as hi / lo pairs (the proj / fc2 path)."""
    prev = lib.vitvs_op_plan_in_flight(plan)
    try:
        g = torch.Generator().manual_seed(M + D + K)
        A = _mk((M, K), g)
        W = _mk((D, K), g, K ** -0.5)
        bias = _mk((D,), g, 0.1)
        ls = (1.0 + 0.3 * _mk((D,), g)) if use_ls else None
        gamma, beta = 1.0 + 0.1 * _mk((D,), g), 0.1 * _mk((D,), g)
        x0 = _mk((M, D), g)
        upd = A.double() @ W.double().t() + bias.double()
        if use_ls:
            upd = upd * ls.double()
        x_ref = x0.double() + upd
        y_ref = torch.nn.functional.layer_norm(x_ref, (D,), gamma.double(), beta.double(), 1e-6)
        slices = lib.vitvs_op_splitk_slices(X2, M, D, K)
        assert 1 <= slices <= 8 and (plan == 1 or slices <= 2)
        e = weight_exp(W)
        Ad, Wd, bd = to_x2(A).cuda(), to_x2(W, e).cuda(), bias.cuda()
        lsd = ls.cuda() if use_ls else None
        gd, bed = gamma.cuda(), beta.cuda()
        x = x0.clone().cuda()
        part = torch.full((slices, M, D), float("nan"), dtype=torch.float32, device="cuda")
        out = torch.full((M, 2 * D), float("nan"), dtype=torch.float16, device="cuda")
        lib.vitvs_op_weight_exponent(e)
        rc = lib.vitvs_op_linear_partial(X2, _p(Ad), _p(Wd), _p(part), M, D, K, slices, _stream())
        lib.vitvs_op_weight_exponent(0)
        assert rc == 0
        assert lib.vitvs_op_residual_ln(X2, _p(x), _p(part), slices, _p(bd), _p(lsd), _p(gd) if use_ln else None,
                                        _p(bed) if use_ln else None, _p(out) if use_ln else None, M, D, 1e-6, _stream()) == 0
        torch.cuda.synchronize()
        ks = K // slices
        for z in range(slices):                              # each slice holds exactly the products of its K range
            ref_z = A[:, z * ks:(z + 1) * ks].double() @ W[:, z * ks:(z + 1) * ks].double().t()
            assert _rel(part[z].cpu(), ref_z) <= TOL
        assert _rel(x.cpu(), x_ref) <= TOL
        if use_ln:
            assert _rel(from_x2(out.cpu()), y_ref) <= TOL
    finally:
        lib.vitvs_op_plan_in_flight(prev)


@pytest.mark.parametrize("M,N,K,use_ls", [(394, 768, 768, False), (130, 384, 1536, True), (1576, 768, 192, True)])
def test_linear_residual(lib, M, N, K, use_ls):
    g = torch.Generator().manual_seed(M + N + K)
    A = _mk((M, K), g)
    W = _mk((N, K), g, K ** -0.5)
    bias = _mk((N,), g, 0.1)
    ls = (1.0 + 0.3 * _mk((N,), g)) if use_ls else None
    x0 = _mk((M, N), g)
    upd = A.double() @ W.double().t() + bias.double()
    if use_ls:
        upd = upd * ls.double()
    ref = x0.double() + upd
    x = x0.clone().cuda()
    e = weight_exp(W)
    Ad, Wd, bd, lsd = to_x2(A).cuda(), to_x2(W, e).cuda(), bias.cuda(), (ls.cuda() if use_ls else None)
    lib.vitvs_op_weight_exponent(e)
    rc = lib.vitvs_op_linear_residual(X2, _p(Ad), _p(Wd), _p(bd), _p(lsd), _p(x), M, N, K, _stream())
    lib.vitvs_op_weight_exponent(0)
    assert rc == 0
    torch.cuda.synchronize()
    assert _rel(x.cpu(), ref) <= TOL


@pytest.mark.parametrize("M,D", [(394, 768), (394, 384), (7, 1024), (1, 128)])
def test_layernorm(lib, M, D):
    g = torch.Generator().manual_seed(D + M)
    x = _mk((M, D), g, 3.0) + 0.7
    gamma = 1.0 + 0.1 * _mk((D,), g)
    beta = 0.1 * _mk((D,), g)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-6)
    out = torch.empty((M, 2 * D), dtype=torch.float16, device="cuda")
    xd, gd, bd = x.cuda(), gamma.cuda(), beta.cuda()
    assert lib.vitvs_op_layernorm(X2, _p(xd), _p(gd), _p(bd), _p(out), M, D, 1e-6, _stream()) == 0
    torch.cuda.synchronize()
    assert _rel(from_x2(out.cpu()), ref) <= 5e-6


def _attention_ref(qkv, n_img, N, H):
    D = H * 64
    q, k, v = qkv.double().reshape(n_img, N, 3, H, 64).unbind(2)
    q, k, v = (t.transpose(1, 2) for t in (q, k, v))
    att = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    return (att @ v).transpose(1, 2).reshape(n_img * N, D)


@pytest.mark.parametrize("n_img,N,H,scale", [(2, 197, 6, 1.0), (2, 197, 12, 3.0), (1, 64, 2, 1.0), (1, 70, 1, 3.0), (2, 257, 2, 2.0),
                                             (1, 1370, 2, 1.0), (3, 5, 1, 1.0), (2, 530, 3, 2.0), (1, 700, 1, 4.0),
                                             (1, 3137, 1, 1.0), (9, 197, 12, 1.0), (16, 485, 6, 1.0),
                                             # from 2048 tokens on: the 128-query kernel (tails of 1 key / 1 query row, several heads and images)
                                             (2, 2049, 2, 2.0), (1, 2200, 3, 1.0), (2, 2048, 1, 3.0)])
def test_attention(lib, n_img, N, H, scale):
    g = torch.Generator().manual_seed(N * 3 + H)
    D = H * 64
    qkv = _mk((n_img * N, 3 * D), g, scale)
    ref = _attention_ref(qkv, n_img, N, H)
    out = torch.full((n_img * N, 2 * D), float("nan"), dtype=torch.float16, device="cuda")
    qd = to_x2(qkv).cuda()
    assert lib.vitvs_op_attention(X2, _p(qd), _p(out), n_img, N, H, _stream()) == 0
    torch.cuda.synchronize()
    got = from_x2(out.cpu())
    assert torch.isfinite(got).all()
    assert _rel(got, ref) <= 1e-5


@pytest.mark.parametrize("N", [130, 2100])
def test_attention_asymmetric_values_catch_transposed_operands(lib, N):
    """V with a per-dimension ramp and one-hot attention: any key / dim permutation slip in the PV product (transposed LDS reads
    through the window swizzle, permuted MFMA k-slots, the hi / lo windows) shows up as an O(1) error.  (130 tokens: the 16-query
    kernel; 2100: the 128-query one.)"""
    H = 1
    qkv = torch.zeros((N, 192), dtype=torch.float32)
    keys = torch.arange(N, dtype=torch.float32)
    qkv[:, 0] = 40.0
    qkv[:, 64] = torch.where(keys == 77, 40.0, -40.0)
    qkv[:, 128:192] = keys[:, None] * 0.5 + torch.arange(64, dtype=torch.float32)[None, :] * 0.01
    ref = _attention_ref(qkv, 1, N, H)
    out = torch.empty((N, 128), dtype=torch.float16, device="cuda")
    qd = to_x2(qkv).cuda()
    assert lib.vitvs_op_attention(X2, _p(qd), _p(out), 1, N, H, _stream()) == 0
    torch.cuda.synchronize()
    assert _rel(from_x2(out.cpu()), ref) <= 1e-5
    assert float(ref[0, 0]) == pytest.approx(38.5, abs=1e-6)


def test_attention_peaky_logits(lib):
    """Logits of tens of units (trained-like q / k gains): where the 16-bit modes lose their arg-maxes."""
    g = torch.Generator().manual_seed(9)
    n_img, N, H = 2, 197, 4
    qkv = _mk((n_img * N, 3 * H * 64), g)
    qkv[:, : 2 * H * 64] *= 4.0                               # |q . k| * 0.125 ~ 16 sigma-units
    ref = _attention_ref(qkv, n_img, N, H)
    out = torch.empty((n_img * N, 2 * H * 64), dtype=torch.float16, device="cuda")
    qd = to_x2(qkv).cuda()
    assert lib.vitvs_op_attention(X2, _p(qd), _p(out), n_img, N, H, _stream()) == 0
    torch.cuda.synchronize()
    assert _rel(from_x2(out.cpu()), ref) <= 2e-5



@pytest.mark.parametrize("variant", [256, 192, 128, 1192, 1256, 1])
@pytest.mark.parametrize("M,N,K,gelu", [(6274, 2304, 768, 0), (2740, 1024, 1024, 1), (1025, 256, 64, 0), (3152, 3072, 768, 1), (300, 512, 192, 0)])
def test_linear_tile_families(lib, variant, M, N, K, gelu):
    """The 256- / 192-row tiles of gemm_big.hip on hi / lo operands (three MFMAs per quadrant k-step, the [hi | lo] epilogue image,
    weights carrying 2^e) and the tiles of gemm.hip on the same shapes: ragged last row tile, the shortest k-loop the big kernel
    takes (K = 64: four k-tiles of 32 logical k ... the kernel's two-tile minimum is K = 64), a row count below one tile."""
    if variant >= 128 and N % {1192: 128, 1256: 256}.get(variant, variant) != 0:
        pytest.skip("this tile width does not divide N")
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = _mk((M, K), g)
    W = _mk((N, K), g, K ** -0.5)
    bias = _mk((N,), g, 0.1)
    ref = A.double() @ W.double().t() + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    e = weight_exp(W)
    Ad, Wd, bd = to_x2(A).cuda(), to_x2(W, e).cuda(), bias.cuda()
    out = torch.full((M + 3, 2 * N), float("nan"), dtype=torch.float16, device="cuda")      # 3 guard rows behind the matrix
    lib.vitvs_op_weight_exponent(e)
    rc = lib.vitvs_op_linear_variant(X2, variant, _p(Ad), _p(Wd), _p(bd), _p(out), M, N, K, gelu, 0, _stream())
    lib.vitvs_op_weight_exponent(0)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isnan(out[M:].float()).all(), "rows beyond M were written"
    got = from_x2(out[:M].cpu())
    assert torch.isfinite(got).all()
    assert _rel(got, ref) <= TOL


@pytest.mark.parametrize("variant", [256, 192, 128, 1192, 1256, 1])
@pytest.mark.parametrize("M,N,K,slices", [(6274, 768, 3072, 3), (2740, 1024, 1024, 2), (1500, 256, 768, 1), (1500, 384, 768, 2)])
def test_linear_partial_tile_families(lib, variant, M, N, K, slices):
    """Split-K partial sums from both tile families: slice z holds exactly the products of its K range (fp32-class)."""
    if variant >= 128 and N % {1192: 128, 1256: 256}.get(variant, variant) != 0:
        pytest.skip("this tile width does not divide N")
    g = torch.Generator().manual_seed(M + N + K + slices)
    A = _mk((M, K), g)
    W = _mk((N, K), g, K ** -0.5)
    e = weight_exp(W)
    Ad, Wd = to_x2(A).cuda(), to_x2(W, e).cuda()
    part = torch.full((slices, M, N), float("nan"), dtype=torch.float32, device="cuda")
    lib.vitvs_op_weight_exponent(e)
    rc = lib.vitvs_op_linear_variant(X2, variant, _p(Ad), _p(Wd), None, _p(part), M, N, K, 0, slices, _stream())
    lib.vitvs_op_weight_exponent(0)
    assert rc == 0
    torch.cuda.synchronize()
    ks = K // slices
    for z in range(slices):
        ref = A[:, z * ks:(z + 1) * ks].double() @ W[:, z * ks:(z + 1) * ks].double().t()
        assert _rel(part[z].cpu(), ref) <= TOL


def test_linear_saturates_instead_of_overflowing(lib):
    """Activations beyond the fp16 range: the hi half saturates at 65504 and the lo half carries the remainder (common.h split4), so
    magnitudes up to 2 x 65504 stay finite and close (the plain fp16 mode turns them into infinities)."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 64, 64, 64
    A = _mk((M, K), g)
    A[3, 5], A[10, 20], A[40, 63] = 9.0e4, -1.2e5, 7.0e4
    W = _mk((N, K), g, K ** -0.5)
    ref = A.double() @ W.double().t()
    e = weight_exp(W)
    Ad, Wd = to_x2(A).cuda(), to_x2(W, e).cuda()
    assert torch.isfinite(Ad.float()).all()
    out = torch.empty((M, 2 * N), dtype=torch.float16, device="cuda")
    bias = torch.zeros(N, device="cuda")
    lib.vitvs_op_weight_exponent(e)
    rc = lib.vitvs_op_linear(X2, _p(Ad), _p(Wd), _p(bias), _p(out), M, N, K, 0, _stream())
    lib.vitvs_op_weight_exponent(0)
    assert rc == 0
    torch.cuda.synchronize()
    got = from_x2(out.cpu())
    assert torch.isfinite(got).all()
    assert _rel(got, ref) <= 2e-4          # the remainder 90000 - 65504 = 24496 has its own 11 bits: 2^-11 * 24496 / 90000
