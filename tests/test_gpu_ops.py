"""Kernel-level parity (GPU): every HIP operator of the forward, driven through the C ABI
(include/vitvs_ops.h), against a plain PyTorch fp32/fp64 statement of the same op.

Tolerances: fp32 kernels run exact-fp32 MFMA chains, so they must agree with an fp64 reference
to fp32 rounding (<= 2e-5 relative to the output scale); bf16 kernels are compared with an fp64
reference evaluated on the SAME bf16-rounded inputs (<= 1.5e-2, the output rounding)."""
import ctypes as C

import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import _lib

pytestmark = pytest.mark.gpu

F32_TOL = 2e-5
BF16_TOL = 1.5e-2


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


F16_TOL = 2e-3   # fp16 operands: 11-bit significand, output rounding 2^-11
PRECS = [("fp32", _lib.F32, torch.float32, F32_TOL), ("bf16", _lib.BF16, torch.bfloat16, BF16_TOL),
         ("fp16", _lib.F16, torch.float16, F16_TOL)]


@pytest.mark.parametrize("name,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("M,N,K,gelu", [(394, 384, 128, 0), (394, 2304, 768, 0), (77, 512, 192, 1), (1200, 3072, 768, 1),
                                        (64, 64, 64, 0), (5, 128, 640, 1), (6274, 3072, 768, 1), (3170, 2304, 768, 0),
                                        (985, 2304, 768, 0), (2364, 3072, 768, 1), (788, 3072, 768, 1)])
def test_linear(lib, name, prec, dtype, tol, M, N, K, gelu):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = _mk((M, K), g).to(dtype)
    W = _mk((N, K), g, K ** -0.5).to(dtype)
    bias = _mk((N,), g, 0.1)
    ref = A.double() @ W.double().t() + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    out = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
    rc = lib.vitvs_op_linear(prec, _p(Ad), _p(Wd), _p(bd), _p(out), M, N, K, gelu, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert _rel(out.cpu(), ref) <= tol


@pytest.fixture
def in_flight_plan(lib):
    """The plan a handle with the "in_flight" option uses (vitvs_op_plan_in_flight): 4-wave workgroups, at most two K slices, the
    XCD-mapped grid of the two-slice partial-sum launches."""
    prev = lib.vitvs_op_plan_in_flight(3)
    yield
    lib.vitvs_op_plan_in_flight(prev)


@pytest.mark.parametrize("name,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("M,N,K,gelu", [(394, 2304, 768, 0), (394, 3072, 768, 1), (197, 1152, 384, 0), (61, 768, 768, 1),
                                        (394, 4096, 1024, 1)])
def test_linear_under_the_in_flight_plan(lib, in_flight_plan, name, prec, dtype, tol, M, N, K, gelu):
    """One-round launches on 4-wave workgroups (64 x 64, 64 x 96, 64 x 128 column tiles), whole-row epilogue, ragged last row tile."""
    t = (C.c_int32 * 3)()
    assert lib.vitvs_op_linear_tile(prec, M, N, K, 0, t) == 0 and t[0] == 64 and t[2] == 1
    g = torch.Generator().manual_seed(M * 7 + N)
    A = _mk((M, K), g).to(dtype)
    W = _mk((N, K), g, K ** -0.5).to(dtype)
    bias = _mk((N,), g, 0.1)
    ref = A.double() @ W.double().t() + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    guard = 3                                            # rows behind the matrix must stay untouched by the row stores
    out = torch.full((M + guard, N), float("nan"), dtype=dtype, device="cuda")
    assert lib.vitvs_op_linear(prec, _p(A.cuda()), _p(W.cuda()), _p(bias.cuda()), _p(out), M, N, K, gelu, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.isnan(out[M:].float()).all()
    assert _rel(out[:M].cpu(), ref) <= tol


@pytest.mark.parametrize("name,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("M,D,K", [(394, 768, 768), (394, 768, 3072), (197, 1024, 4096), (61, 768, 3072), (394, 384, 1536)])
def test_split_k_pair_under_the_in_flight_plan(lib, in_flight_plan, name, prec, dtype, tol, M, D, K):
    """Two K slices on the XCD-mapped 1-D grid (D / 64 column tiles a multiple of 4; 384 / 64 = 6 keeps the plain grid), summed by
    residual_ln: equal to the reference, and each slice holds the sum over ITS half of K whichever workgroup computed it — with
    the map and without."""
    slices = lib.vitvs_op_splitk_slices(prec, M, D, K)
    assert slices == 2
    g = torch.Generator().manual_seed(M + D + K)
    A = _mk((M, K), g).to(dtype)
    W = _mk((D, K), g, K ** -0.5).to(dtype)
    bias = _mk((D,), g, 0.1)
    x0 = _mk((M, D), g)
    x_ref = x0.double() + A.double() @ W.double().t() + bias.double()
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    x = x0.clone().cuda()
    part = torch.full((slices, M, D), float("nan"), dtype=torch.float32, device="cuda")
    assert lib.vitvs_op_linear_partial(prec, _p(Ad), _p(Wd), _p(part), M, D, K, slices, _stream()) == 0
    assert lib.vitvs_op_residual_ln(prec, _p(x), _p(part), slices, _p(bd), None, None, None, None, M, D, 1e-6, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(part).all()
    assert _rel(x.cpu(), x_ref) <= (tol if prec == _lib.F32 else 2e-6 + 1e-3)
    # the same two slices without the hint (plain 3-D grid, the 8-wave plan where it applies)
    lib.vitvs_op_plan_in_flight(1)
    part1 = torch.full((slices, M, D), float("nan"), dtype=torch.float32, device="cuda")
    assert lib.vitvs_op_linear_partial(prec, _p(Ad), _p(Wd), _p(part1), M, D, K, slices, _stream()) == 0
    torch.cuda.synchronize()
    lib.vitvs_op_plan_in_flight(3)
    ref_slices = torch.stack([A[:, z * (K // 2):(z + 1) * (K // 2)].double() @ W[:, z * (K // 2):(z + 1) * (K // 2)].double().t()
                              for z in range(2)])
    for got in (part, part1):                            # each slice holds ITS half of K, whichever workgroup computed it
        assert _rel(got.cpu(), ref_slices) <= (tol if prec == _lib.F32 else 1e-3)


@pytest.mark.parametrize("name,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("M,N,K,use_ls", [(394, 768, 768, False), (394, 768, 3072, True), (130, 384, 1536, True),
                                          # 300 / 600 workgroups of 64 x 64: the 3- and the 2-stage ring (ring_stages, gemm.hip);
                                          # K = 192 / 64: fewer k-tiles than the ring has stages
                                          (1576, 768, 768, True), (3152, 768, 768, False), (1576, 768, 192, False),
                                          (3100, 768, 64, True)])
def test_linear_residual(lib, name, prec, dtype, tol, M, N, K, use_ls):
    g = torch.Generator().manual_seed(M + N + K)
    A = _mk((M, K), g).to(dtype)
    W = _mk((N, K), g, K ** -0.5).to(dtype)
    bias = _mk((N,), g, 0.1)
    ls = (1.0 + 0.3 * _mk((N,), g)) if use_ls else None
    x0 = _mk((M, N), g)
    upd = A.double() @ W.double().t() + bias.double()
    if use_ls:
        upd = upd * ls.double()
    ref = x0.double() + upd
    x = x0.clone().cuda()
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    lsd = ls.cuda() if use_ls else None
    rc = lib.vitvs_op_linear_residual(prec, _p(Ad), _p(Wd), _p(bd), _p(lsd), _p(x), M, N, K, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert _rel(x.cpu(), ref) <= (tol if prec == _lib.F32 else 2e-6 + 1e-3)  # 16-bit operands: fp32 accumulate, fp32 output


@pytest.mark.parametrize("name,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("M,D", [(394, 768), (394, 384), (7, 1024), (1, 128), (2740, 1024)])
def test_layernorm(lib, name, prec, dtype, tol, M, D):
    g = torch.Generator().manual_seed(D + M)
    x = _mk((M, D), g, 3.0) + 0.7
    gamma = 1.0 + 0.1 * _mk((D,), g)
    beta = 0.1 * _mk((D,), g)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-6)
    out = torch.empty((M, D), dtype=dtype, device="cuda")
    xd, gd, bd = x.cuda(), gamma.cuda(), beta.cuda()
    rc = lib.vitvs_op_layernorm(prec, _p(xd), _p(gd), _p(bd), _p(out), M, D, 1e-6, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert _rel(out.cpu(), ref) <= {_lib.F32: 5e-6, _lib.BF16: 8e-3, _lib.F16: 1e-3}[prec]


def _attention_ref(qkv, n_img, N, H):
    D = H * 64
    q, k, v = qkv.double().reshape(n_img, N, 3, H, 64).unbind(2)
    q, k, v = (t.transpose(1, 2) for t in (q, k, v))
    att = ((q @ k.transpose(-2, -1)) * 0.125).softmax(-1)
    return (att @ v).transpose(1, 2).reshape(n_img * N, D)


@pytest.mark.parametrize("name,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("n_img,N,H,scale", [(2, 197, 6, 1.0), (1, 64, 2, 1.0), (1, 70, 1, 3.0), (2, 257, 2, 2.0),
                                             (1, 1370, 2, 1.0), (3, 5, 1, 1.0), (2, 530, 3, 2.0), (1, 700, 1, 4.0),
                                             (1, 3137, 1, 1.0), (9, 197, 12, 1.0), (16, 197, 12, 1.0), (24, 130, 6, 2.0),
                                             (48, 128, 4, 1.0), (7, 128, 8, 1.0), (16, 485, 6, 1.0)])
def test_attention(lib, name, prec, dtype, tol, n_img, N, H, scale):
    g = torch.Generator().manual_seed(N * 3 + H)
    D = H * 64
    qkv = _mk((n_img * N, 3 * D), g, scale).to(dtype)
    ref = _attention_ref(qkv, n_img, N, H)
    out = torch.full((n_img * N, D), float("nan"), dtype=dtype, device="cuda")
    qd = qkv.cuda()
    rc = lib.vitvs_op_attention(prec, _p(qd), _p(out), n_img, N, H, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    # raw q: from 512 tokens on the 16-bit kernel scales q itself, one more 16-bit rounding of q than the forward's form below
    assert _rel(out.cpu(), ref) <= {_lib.F32: 1e-5, _lib.BF16: 2e-2, _lib.F16: 4e-3 if N >= 512 else 3e-3}[prec]
    if prec != _lib.F32:
        # the forward's form: q carries 0.125 * log2(e), applied in fp32 BEFORE the one rounding to 16 bits (the handle folds
        # it into the q rows of the qkv weights); reference = softmax of the values the kernel is given, 2^(q' . k)
        qs = qkv.clone().float()
        qs[:, :D] *= 0.125 * 1.4426950408889634
        qs = qs.to(dtype)
        t = qs.double().clone()
        t[:, :D] /= 0.125 * 1.4426950408889634
        ref_q = _attention_ref(t, n_img, N, H)
        out2 = torch.full((n_img * N, D), float("nan"), dtype=dtype, device="cuda")
        qsd = qs.cuda()
        assert lib.vitvs_op_attention_q(prec, _p(qsd), _p(out2), n_img, N, H, 1, _stream()) == 0
        torch.cuda.synchronize()
        assert _rel(out2.cpu(), ref_q) <= {_lib.BF16: 2e-2, _lib.F16: 3e-3}[prec]


def test_attention_asymmetric_values_catch_transposed_operands(lib):
    """V with a per-dimension ramp and one-hot attention: any key/dim permutation slip in the PV
    product (transposed LDS reads, permuted MFMA k-slots) shows up as an O(1) error."""
    N, H = 130, 1
    qkv = torch.zeros((N, 192), dtype=torch.float32)
    keys = torch.arange(N, dtype=torch.float32)
    qkv[:, 0] = 40.0  # q . k = 40 * k[0]
    qkv[:, 64] = torch.where(keys == 77, 40.0, -40.0)  # softmax collapses onto key 77
    qkv[:, 128:192] = keys[:, None] * 0.5 + torch.arange(64, dtype=torch.float32)[None, :] * 0.01
    ref = _attention_ref(qkv, 1, N, H)
    for prec, dtype, tol in ((_lib.F32, torch.float32, 1e-5), (_lib.BF16, torch.bfloat16, 1e-2), (_lib.F16, torch.float16, 2e-3)):
        q = qkv.to(dtype)
        out = torch.empty((N, 64), dtype=dtype, device="cuda")
        qd = q.cuda()
        assert lib.vitvs_op_attention(prec, _p(qd), _p(out), 1, N, H, _stream()) == 0
        torch.cuda.synchronize()
        assert _rel(out.cpu(), _attention_ref(q, 1, N, H)) <= tol
    assert float(ref[0, 0]) == pytest.approx(38.5, abs=1e-6)


@pytest.mark.parametrize("name,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("M,D,K,use_ls,use_ln", [(394, 768, 768, False, True), (394, 768, 3072, True, True),
                                                  (130, 384, 1536, False, False), (6274, 768, 768, False, True),
                                                  (3152, 768, 3072, False, True), (2740, 1024, 4096, True, True),
                                                  (2740, 1024, 1024, False, True)])
def test_split_k_pair(lib, name, prec, dtype, tol, M, D, K, use_ls, use_ln):
    """linear_partial + residual_ln == x + ls * (A W^T + bias), then LayerNorm (the proj / fc2 path of the forward)."""
    g = torch.Generator().manual_seed(M + D + K)
    A = _mk((M, K), g).to(dtype)
    W = _mk((D, K), g, K ** -0.5).to(dtype)
    bias = _mk((D,), g, 0.1)
    ls = (1.0 + 0.3 * _mk((D,), g)) if use_ls else None
    gamma, beta = 1.0 + 0.1 * _mk((D,), g), 0.1 * _mk((D,), g)
    x0 = _mk((M, D), g)
    upd = A.double() @ W.double().t() + bias.double()
    if use_ls:
        upd = upd * ls.double()
    x_ref = x0.double() + upd
    y_ref = torch.nn.functional.layer_norm(x_ref, (D,), gamma.double(), beta.double(), 1e-6)
    slices = lib.vitvs_op_splitk_slices(prec, M, D, K)
    assert 1 <= slices <= 8
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    lsd = ls.cuda() if use_ls else None
    gd, bed = gamma.cuda(), beta.cuda()
    x = x0.clone().cuda()
    part = torch.full((slices, M, D), float("nan"), dtype=torch.float32, device="cuda")
    out = torch.full((M, D), float("nan"), dtype=dtype, device="cuda")
    assert lib.vitvs_op_linear_partial(prec, _p(Ad), _p(Wd), _p(part), M, D, K, slices, _stream()) == 0
    assert lib.vitvs_op_residual_ln(prec, _p(x), _p(part), slices, _p(bd), _p(lsd), _p(gd) if use_ln else None,
                                    _p(bed) if use_ln else None, _p(out) if use_ln else None, M, D, 1e-6, _stream()) == 0
    torch.cuda.synchronize()
    assert _rel(x.cpu(), x_ref) <= (tol if prec == _lib.F32 else 2e-6 + 1e-3)      # fp32 accumulate, fp32 residual stream
    if use_ln:
        assert _rel(out.cpu(), y_ref) <= {_lib.F32: 2e-5, _lib.BF16: 1e-2, _lib.F16: 2e-3}[prec]


@pytest.mark.parametrize("name,prec,dtype,tol", [p for p in PRECS if p[0] != "fp32"])
@pytest.mark.parametrize("variant", [256, 192, 128, 1192, 1256, 1, 2])
@pytest.mark.parametrize("M,N,K,gelu", [(6274, 2304, 768, 0), (2740, 1024, 1024, 1), (1025, 256, 128, 0), (3152, 768, 3072, 1),
                                        (300, 512, 192, 0), (1025, 384, 128, 1), (3152, 3072, 768, 1)])
def test_linear_tile_families_agree_with_the_reference(lib, name, prec, dtype, tol, variant, M, N, K, gelu):
    """The 256-row tiles of gemm_big.hip (both column widths) and the tiles of gemm.hip on the same shapes: ragged last
    row tile (M % 256 != 0), the shortest k-loop the big kernel accepts (K = 128: prologue + two tail k-tiles only), odd
    and even k-tile counts, and a row count below one tile."""
    if variant >= 128 and N % {1192: 128, 1256: 256}.get(variant, variant) != 0:
        pytest.skip("this tile width does not divide N")
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = _mk((M, K), g).to(dtype)
    W = _mk((N, K), g, K ** -0.5).to(dtype)
    bias = _mk((N,), g, 0.1)
    ref = A.double() @ W.double().t() + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    out = torch.full((M + 3, N), float("nan"), dtype=dtype, device="cuda")      # 3 guard rows behind the matrix
    rc = lib.vitvs_op_linear_variant(prec, variant, _p(Ad), _p(Wd), _p(bd), _p(out), M, N, K, gelu, 0, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isnan(out[M:].float()).all(), "rows beyond M were written"
    assert torch.isfinite(out[:M].float()).all()
    assert _rel(out[:M].cpu(), ref) <= tol


@pytest.mark.parametrize("variant", [256, 192, 128, 1192, 1256, 1, 2])
@pytest.mark.parametrize("M,N,K,slices", [(6274, 768, 3072, 3), (2740, 1024, 1024, 2), (1500, 256, 768, 1), (1500, 384, 768, 2)])
def test_linear_partial_tile_families(lib, variant, M, N, K, slices):
    """Split-K partial sums from both tile families: slice z holds exactly the products of its K range (fp32)."""
    if variant >= 128 and N % {1192: 128, 1256: 256}.get(variant, variant) != 0:
        pytest.skip("this tile width does not divide N")
    g = torch.Generator().manual_seed(M + N + K + slices)
    A = _mk((M, K), g).to(torch.bfloat16)
    W = _mk((N, K), g, K ** -0.5).to(torch.bfloat16)
    Ad, Wd = A.cuda(), W.cuda()
    part = torch.full((slices, M, N), float("nan"), dtype=torch.float32, device="cuda")
    rc = lib.vitvs_op_linear_variant(_lib.BF16, variant, _p(Ad), _p(Wd), None, _p(part), M, N, K, 0, slices, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    ks = K // slices
    for z in range(slices):
        ref = A[:, z * ks:(z + 1) * ks].double() @ W[:, z * ks:(z + 1) * ks].double().t()
        assert _rel(part[z].cpu(), ref) <= 2e-6 + 1e-3


def test_long_attention_rescale_branch_is_forced(lib):
    """Online softmax of the 128-query kernel (N >= 512): the running maximum of chosen queries jumps at chosen key tiles
    (a key that matches them far better than every earlier one), so the exact-rescale branch is taken late in the key
    loop, with and without a ragged last tile; full-tensor fp64 reference."""
    for N in (640, 777):
        g = torch.Generator().manual_seed(N)
        qkv = _mk((N, 192), g, 0.5)
        for qrow, krow in ((3, 70), (130, 400), (131, 639), (600, N - 1), (17, 5)):
            qkv[krow, 64:128] = qkv[qrow, 0:64] * 6.0          # key `krow` aligned with query `qrow`: a late, large maximum
        for prec, dtype, tol in ((_lib.BF16, torch.bfloat16, 2e-2), (_lib.F16, torch.float16, 3e-3)):
            q = qkv.to(dtype)
            out = torch.full((N, 64), float("nan"), dtype=dtype, device="cuda")
            qd = q.cuda()
            assert lib.vitvs_op_attention(prec, _p(qd), _p(out), 1, N, 1, _stream()) == 0
            torch.cuda.synchronize()
            assert _rel(out.cpu(), _attention_ref(q, 1, N, 1)) <= tol


def test_split_attention_on_two_streams_at_once(lib):
    """The key-split attention keeps per-launch state (partial softmax states, tickets) in a workspace.  The operator hook's
    workspace is per (device, stream): two streams running split launches concurrently on different inputs must each get
    the result they get alone (bit for bit), launch after launch."""
    n_img, N, H = 1, 1370, 16                                   # 352 query blocks -> keys split in two ranges
    D = H * 64
    g = torch.Generator().manual_seed(99)
    qa = _mk((n_img * N, 3 * D), g).to(torch.bfloat16).cuda()
    qb = _mk((n_img * N, 3 * D), g, 2.0).to(torch.bfloat16).cuda()
    alone = []
    for q in (qa, qb):
        out = torch.empty((n_img * N, D), dtype=torch.bfloat16, device="cuda")
        assert lib.vitvs_op_attention(_lib.BF16, _p(q), _p(out), n_img, N, H, _stream()) == 0
        torch.cuda.synchronize()
        alone.append(out.clone())
    assert _rel(alone[0].cpu(), _attention_ref(qa.cpu(), n_img, N, H)) <= 2e-2
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a = [torch.empty_like(alone[0]) for _ in range(6)]
    outs_b = [torch.empty_like(alone[1]) for _ in range(6)]
    torch.cuda.synchronize()
    for i in range(6):                                          # interleaved enqueue: the launches of the two streams overlap
        assert lib.vitvs_op_attention(_lib.BF16, _p(qa), _p(outs_a[i]), n_img, N, H, C.c_void_p(sa.cuda_stream)) == 0
        assert lib.vitvs_op_attention(_lib.BF16, _p(qb), _p(outs_b[i]), n_img, N, H, C.c_void_p(sb.cuda_stream)) == 0
    torch.cuda.synchronize()
    for i in range(6):
        assert torch.equal(outs_a[i], alone[0]) and torch.equal(outs_b[i], alone[1])
    # ONE stream, alternating inputs back to back: every launch reuses the same workspace slots for different states; a merge
    # that read a range's state from an older launch (a stale cache line behind the fence-free hand-off) would mix A and B
    seq = [qa, qb, qb, qa, qb, qa, qa, qb]
    outs = [torch.empty_like(alone[0]) for _ in seq]
    for q, o in zip(seq, outs):
        assert lib.vitvs_op_attention(_lib.BF16, _p(q), _p(o), n_img, N, H, _stream()) == 0
    torch.cuda.synchronize()
    for q, o in zip(seq, outs):
        assert torch.equal(o, alone[0] if q is qa else alone[1])


@pytest.mark.parametrize("slope", [0.004, 0.02, 0.05, 0.2])
def test_long_attention_deferred_rescale_with_slowly_growing_scores(lib, slope):
    """The 128-query kernel subtracts an EARLIER tile's maximum inside the score MFMAs and accepts a tile without forming its
    maximum while every exp2 stays <= 64 (attention.hip).  Scores that grow steadily along the key axis keep that path busy:
    per 64-key tile the maximum rises by 0.4 / 1.8 / 4.6 / 18 log2 units (slope per key in natural units), i.e. the shift lags
    for many tiles / a few tiles / one tile / never.  P values up to 64 enter the PV product and the running sums; the
    result must still be the softmax, against a full-tensor fp64 reference (CDNA4 guide T13: test the deferral itself)."""
    N, H = 1100, 2
    g = torch.Generator().manual_seed(int(slope * 1000))
    qkv = _mk((N, 3 * 64 * H), g, 0.3)
    keys = torch.arange(N, dtype=torch.float32)
    for h in range(H):
        qkv[:, h * 64] = 8.0                                            # q . k / 8 = k[0] + noise
        qkv[:, 128 + h * 64] = keys * slope * (1.0 if h == 0 else -1.0)  # head 0: growing, head 1: falling (never rescales)
    for prec, dtype, tol in ((_lib.BF16, torch.bfloat16, 2e-2), (_lib.F16, torch.float16, 3e-3)):
        q = qkv.to(dtype)
        out = torch.full((N, 64 * H), float("nan"), dtype=dtype, device="cuda")
        qd = q.cuda()
        assert lib.vitvs_op_attention(prec, _p(qd), _p(out), 1, N, H, _stream()) == 0
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        assert _rel(out.cpu(), _attention_ref(q, 1, N, H)) <= tol


# ------------------------------------------------------------------------------------------------------------------
# The many-row selections of the "in_flight" hint (vitvs_set_option; kernels.h g_updates_in_flight): 256 x 256 tiles wherever they
# divide from 1536 rows on (gemm_big.hip big_tile_width), at most two K slices and exactly two from 2048 rows on for the narrow
# layers (gemm.hip splitk_slices), whole query blocks in the long-sequence attention (attention.hip attention_plan).  Row counts:
# 4 / 8 ViT-B/16 pairs (1576 / 3152), ViT-B/8 448² (6274), ViT-L/14 518² (2740).
@pytest.mark.parametrize("name,prec,dtype,tol", [p for p in PRECS if p[0] != "fp32"])
@pytest.mark.parametrize("M,N,K,gelu", [(1576, 2304, 768, 0), (1576, 3072, 768, 1), (3152, 2304, 768, 0), (3152, 3072, 768, 1),
                                        (6274, 2304, 768, 0), (6274, 3072, 768, 1), (2740, 3072, 1024, 0), (2740, 4096, 1024, 1)])
def test_linear_many_rows_under_the_in_flight_plan(lib, in_flight_plan, name, prec, dtype, tol, M, N, K, gelu):
    t = (C.c_int32 * 3)()
    assert lib.vitvs_op_linear_tile(prec, M, N, K, 0, t) == 0
    assert (t[0], t[1], t[2]) == (256, 256, 0), f"the hint should select 256 x 256 tiles here, got {tuple(t)}"
    lib.vitvs_op_plan_in_flight(1)
    assert lib.vitvs_op_linear_tile(prec, M, N, K, 0, t) == 0
    alone = tuple(t)
    lib.vitvs_op_plan_in_flight(3)
    g = torch.Generator().manual_seed(M * 7 + N)
    A = _mk((M, K), g).to(dtype)
    W = _mk((N, K), g, K ** -0.5).to(dtype)
    bias = _mk((N,), g, 0.1)
    ref = A.double() @ W.double().t() + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    out = torch.full((M + 3, N), float("nan"), dtype=dtype, device="cuda")
    assert lib.vitvs_op_linear(prec, _p(A.cuda()), _p(W.cuda()), _p(bias.cuda()), _p(out), M, N, K, gelu, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.isnan(out[M:].float()).all(), "rows beyond M were written"
    assert _rel(out[:M].cpu(), ref) <= tol, f"(the alone plan's tile here: {alone})"


@pytest.mark.parametrize("name,prec,dtype,tol", [p for p in PRECS if p[0] != "fp32"])
@pytest.mark.parametrize("M,D,K,use_ls", [(1576, 768, 768, False), (1576, 768, 3072, False), (3152, 768, 768, False),
                                          (3152, 768, 3072, False), (6274, 768, 768, False), (6274, 768, 3072, False),
                                          (2740, 1024, 1024, True), (2740, 1024, 4096, True)])
def test_split_k_pair_many_rows_under_the_in_flight_plan(lib, in_flight_plan, name, prec, dtype, tol, M, D, K, use_ls):
    """proj / fc2 at many rows under the hint: the slice count the library picks there (at most 2 from 2048 rows on, where the alone
    plan cuts 3 or 4; below that the many-row rule's 2 .. 4), the partial sums slice by slice, and residual_ln's sum + LayerNorm."""
    slices = lib.vitvs_op_splitk_slices(prec, M, D, K)
    lib.vitvs_op_plan_in_flight(1)
    slices_alone = lib.vitvs_op_splitk_slices(prec, M, D, K)
    lib.vitvs_op_plan_in_flight(3)
    assert 1 <= slices <= (2 if M >= 2048 else 4) and slices <= slices_alone
    g = torch.Generator().manual_seed(M + D + K)
    A = _mk((M, K), g).to(dtype)
    W = _mk((D, K), g, K ** -0.5).to(dtype)
    bias = _mk((D,), g, 0.1)
    ls = (1.0 + 0.3 * _mk((D,), g)) if use_ls else None
    gamma, beta = 1.0 + 0.1 * _mk((D,), g), 0.1 * _mk((D,), g)
    x0 = _mk((M, D), g)
    upd = A.double() @ W.double().t() + bias.double()
    if use_ls:
        upd = upd * ls.double()
    x_ref = x0.double() + upd
    y_ref = torch.nn.functional.layer_norm(x_ref, (D,), gamma.double(), beta.double(), 1e-6)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    lsd = ls.cuda() if use_ls else None
    x = x0.clone().cuda()
    part = torch.full((slices, M, D), float("nan"), dtype=torch.float32, device="cuda")
    out = torch.full((M, D), float("nan"), dtype=dtype, device="cuda")
    assert lib.vitvs_op_linear_partial(prec, _p(Ad), _p(Wd), _p(part), M, D, K, slices, _stream()) == 0
    assert lib.vitvs_op_residual_ln(prec, _p(x), _p(part), slices, _p(bd), _p(lsd), _p(gamma.cuda()), _p(beta.cuda()), _p(out),
                                    M, D, 1e-6, _stream()) == 0
    torch.cuda.synchronize()
    ks = K // slices
    for z in range(slices):
        ref_z = A[:, z * ks:(z + 1) * ks].double() @ W[:, z * ks:(z + 1) * ks].double().t()
        assert _rel(part[z].cpu(), ref_z) <= 1e-3
    assert _rel(x.cpu(), x_ref) <= 2e-6 + 1e-3
    assert _rel(out.cpu(), y_ref) <= {_lib.BF16: 1e-2, _lib.F16: 2e-3}[prec]


@pytest.mark.parametrize("name,prec,dtype,tol", [p for p in PRECS if p[0] != "fp32"])
@pytest.mark.parametrize("n_img,N,H", [(2, 3137, 12), (2, 1370, 16), (1, 3026, 6), (16, 197, 12)])
def test_attention_under_the_in_flight_plan(lib, in_flight_plan, name, prec, dtype, tol, n_img, N, H):
    """Long sequences under the hint: whole query blocks per workgroup (no key ranges, no hand-off), with the forward's
    pre-scaled q; against the fp64 softmax and bit for bit against the divided plan's merge order?  No: the divided plan sums
    the same tiles in ranges, so the two agree to rounding only — asserted at the operator tolerance."""
    g = torch.Generator().manual_seed(N * 3 + H)
    D = H * 64
    qkv = _mk((n_img * N, 3 * D), g).to(dtype)
    qs = qkv.clone().float()
    qs[:, :D] *= 0.125 * 1.4426950408889634
    qs = qs.to(dtype)
    t = qs.double().clone()
    t[:, :D] /= 0.125 * 1.4426950408889634
    ref = _attention_ref(t, n_img, N, H)
    qd = qs.cuda()
    out = torch.full((n_img * N, D), float("nan"), dtype=dtype, device="cuda")
    assert lib.vitvs_op_attention_q(prec, _p(qd), _p(out), n_img, N, H, 1, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    bar = {_lib.BF16: 2e-2, _lib.F16: 3e-3}[prec]
    assert _rel(out.cpu(), ref) <= bar
    lib.vitvs_op_plan_in_flight(1)
    out1 = torch.full((n_img * N, D), float("nan"), dtype=dtype, device="cuda")
    assert lib.vitvs_op_attention_q(prec, _p(qd), _p(out1), n_img, N, H, 1, _stream()) == 0
    torch.cuda.synchronize()
    lib.vitvs_op_plan_in_flight(3)
    assert _rel(out1.cpu(), ref) <= bar and _rel(out.cpu(), out1.cpu()) <= bar


def test_a_16bit_many_token_handle_created_under_a_thread_hint_still_runs_alone(lib):
    """vitvs_create sizes the key-split attention workspace for the ALONE plan whatever vitvs_op_plan_in_flight left on the
    calling thread: a bf16 ViT-B/8 448² handle created under hint 3 and then used with its own in_flight = 1 must
    find its states and tickets (it used to fail every forward with -3)."""
    import numpy as np
    from vitvs_amd import config, synth, weights
    from vitvs_amd.engine import Engine
    prev = lib.vitvs_op_plan_in_flight(3)
    try:
        cfg = config.baseline_config("vitb8_448")
        eng = Engine(cfg, config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False), precision="bf16", max_pairs=1)
    finally:
        lib.vitvs_op_plan_in_flight(prev)
    eng.load_state_dict(weights.synthetic_state_dict(cfg, 0))
    toks = eng.forward_tokens(synth.frame_pair(cfg.img_size, 3)[0][None])
    torch.cuda.synchronize()
    assert torch.isfinite(toks).all() and np.isfinite(float(toks.abs().max()))
    eng.close()
