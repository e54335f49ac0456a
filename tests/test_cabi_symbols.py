"""CPU-side checks of the drop-in boundary: libvitvs_hip.so builds (hipcc cross-compiles gfx950
without a GPU), loads, and exports every symbol include/*.h declares.  No compute calls here."""
import ctypes
import os
import re

import pytest

import vitvs_amd  # noqa: F401
from vitvs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.isfile(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def _declared(header):
    with open(os.path.join(ROOT, "include", header)) as fh:
        text = fh.read()
    return set(re.findall(r"VITVS_API\s+[\w\s\*]+?\b(vitvs_\w+)\s*\(", text))


def test_headers_and_binding_agree(lib):
    declared = _declared("vitvs.h") | _declared("vitvs_ops.h")
    assert declared, "no prototypes found in include/"
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))


def test_every_declared_symbol_is_exported(lib):
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared("vitvs.h") | _declared("vitvs_ops.h"):
        assert hasattr(raw, name), f"{name} not exported"


def test_abi_version_and_config_layout(lib):
    assert lib.vitvs_abi_version() == _lib.ABI_VERSION
    # struct vitvs_config: 8 int32, 7 float, 5 int32, double, 2 int32 -> 96 bytes (static_assert'ed in api.hip)
    assert ctypes.sizeof(_lib.VitvsConfig) == 96
    assert _lib.VitvsConfig.lambda_.offset == 80


def test_create_rejects_bad_config_without_touching_a_gpu(lib):
    cfg = _lib.VitvsConfig()
    cfg.abi_version = 999
    h = ctypes.c_void_p()
    assert lib.vitvs_create(ctypes.byref(cfg), ctypes.byref(h)) < 0
    assert b"abi_version" in lib.vitvs_last_error(None)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from vitvs_amd import config
    from vitvs_amd.engine import Engine, VitvsError
    with pytest.raises(VitvsError):
        Engine(config.baseline_config("vits16_224"))


def test_bench_split_k_mirror_matches_the_library_plan():
    """bench.py prices the split-K GEMM with its own mirror of the slice plan; the plan itself is host arithmetic in the
    library (no device call), so the two can be compared without a GPU."""
    import importlib.util
    import os
    from vitvs_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lib = _lib.load()
    for prec, bk in ((_lib.BF16, 64), (_lib.F16, 64), (_lib.F32, 32)):
        for m in (197, 394, 788, 1576, 3152, 970, 2740, 6274):
            for n, k in ((768, 768), (768, 3072), (384, 384), (384, 1536), (1024, 1024), (1024, 4096), (768, 640), (768, 192)):
                assert bench.split_k(m, n, k, bk) == lib.vitvs_op_splitk_slices(prec, m, n, k), (prec, m, n, k)
    # ... and under the plan hint of a handle whose updates run beside others (vitvs_set_option "in_flight")
    assert lib.vitvs_op_plan_in_flight(3) == 1
    try:
        for m in (197, 394, 788, 1576, 2364, 3152, 2740, 6274):
            for n, k in ((768, 768), (768, 3072), (384, 1536), (1024, 4096)):
                assert bench.split_k(m, n, k, 64, 3) == lib.vitvs_op_splitk_slices(_lib.BF16, m, n, k), (m, n, k)
        t = (ctypes.c_int32 * 3)()
        assert lib.vitvs_op_linear_tile(_lib.BF16, 394, 2304, 768, 0, t) == 0 and list(t) == [64, 64, 1]   # 4-wave workgroups
    finally:
        assert lib.vitvs_op_plan_in_flight(1) == 3
    t = (ctypes.c_int32 * 3)()
    assert lib.vitvs_op_linear_tile(_lib.BF16, 394, 2304, 768, 0, t) == 0 and list(t) == [64, 64, 2]


def test_linear_tile_plan_of_the_library():
    """The tile family per layer shape is host arithmetic (vitvs_op_linear_tile, no device call).  Pins the measured choices:
    64-row tiles while they cover the layer in one round of workgroups, 256-row tiles where they would need a second one, and
    between 256 x 256 and 256 x 128 the one that leaves the busiest CU the least work (profiles/r02_notes.md section 2)."""
    import ctypes
    from vitvs_amd import _lib
    lib = _lib.load()

    def plan(prec, m, n, k, slices=0):
        t = (ctypes.c_int32 * 3)()
        assert lib.vitvs_op_linear_tile(prec, m, n, k, slices, t) == 0
        return tuple(t)
    b = _lib.BF16
    assert plan(b, 394, 3072, 768) == (64, 96, 2)            # one frame pair: fc1, qkv, the K-sliced narrow layers
    assert plan(b, 394, 2304, 768) == (64, 64, 2)
    assert plan(b, 394, 768, 3072, 3) == (64, 64, 2)
    assert plan(b, 788, 2304, 768) == (64, 128, 2)           # two pairs: still one round of 64-row tiles
    assert plan(b, 788, 3072, 768) == (64, 128, 1)           # 312 workgroups on a 2-stage ring (3 per CU) against 120 tiles of 192 x 128
    assert plan(b, 985, 2304, 768) == (64, 128, 1)           # rotation search (5 images): 288 workgroups, still one round
    assert plan(b, 1182, 2304, 768) == (192, 128, 0)         # 342 workgroups: no gain in the forward, the persistent tiles keep it
    assert plan(b, 2364, 3072, 768) == (256, 128, 0)         # 120 tiles of 256 x 256 would leave half the CUs idle
    assert plan(b, 3152, 3072, 768) == (256, 192, 0)         # 8 pairs: 208 tiles of 256 x 192 in one round beat 156 of 256 x 256
    assert plan(b, 3152, 2304, 768) == (256, 128, 0)
    assert plan(b, 3152, 768, 3072, 3) == (256, 128, 0)
    assert plan(b, 3152, 768, 768, 1) == (64, 64, 1)
    assert plan(b, 6274, 2304, 768) == (256, 256, 0)         # ViT-B/8 448
    assert plan(b, 6274, 3072, 768) == (256, 192, 0)         # 400 tiles in two rounds beat 600 of 256 x 128 in three
    assert plan(b, 2740, 3072, 1024) == (256, 192, 0)        # ViT-L/14 518
    assert plan(b, 2740, 4096, 1024) == (192, 256, 0)        # 240 tiles against 176 of 256 x 256
    assert plan(b, 2740, 1024, 4096, 2) == (192, 128, 0)     # 240 tiles of 192 x 128 in one round against 176 of 256 x 128
    assert plan(b, 6274, 768, 3072, 1) == (192, 128, 0)
    assert plan(_lib.F32, 6274, 3072, 768) == (128, 128, 1)  # fp32 never takes the 256-row kernels
    t = (ctypes.c_int32 * 3)()
    assert lib.vitvs_op_linear_tile(b, 394, 768, 3072, 5, t) != 0   # 3072 is not a multiple of 5 k-tiles
