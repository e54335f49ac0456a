"""The resize oracle (oracle/resize_ref.py, a restatement of Pillow's Resample.c) against Pillow itself — the
dependency the reference calls at vitvs_v2.py:474-475.  Bit-identity is required (uint8 work)."""
import numpy as np
import pytest

from oracle import resize_ref

PIL_Image = pytest.importorskip("PIL.Image")

CASES = [(480, 640, 224), (480, 640, 308), (480, 640, 518), (224, 224, 224), (100, 37, 224), (720, 1280, 448), (17, 23, 64)]


@pytest.mark.parametrize("h,w,out", CASES)
def test_restatement_matches_pillow_random(h, w, out):
    rng = np.random.default_rng(h * 1000 + w + out)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    ref = np.asarray(PIL_Image.fromarray(img).resize((out, out)))
    got = resize_ref.resize_bicubic_u8(img, out)
    assert got.dtype == np.uint8 and got.shape == (out, out, 3)
    assert np.array_equal(got, ref)


def test_restatement_matches_pillow_structured():
    """Saturating edges (over/undershoot of the negative lobes must clip exactly like Pillow's clip8 table)."""
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    img[:, ::7] = 255
    img[::5, :, 1] = 255
    img[100:300, 200:400, 2] = 255
    for out in (224, 308):
        assert np.array_equal(resize_ref.resize_bicubic_u8(img, out), np.asarray(PIL_Image.fromarray(img).resize((out, out))))


def test_default_filter_is_bicubic():
    """The reference passes no filter: pin that the default of this Pillow is BICUBIC (what the restatement encodes)."""
    rng = np.random.default_rng(5)
    img = PIL_Image.fromarray(rng.integers(0, 256, size=(48, 64, 3), dtype=np.uint8))
    assert np.array_equal(np.asarray(img.resize((24, 24))), np.asarray(img.resize((24, 24), PIL_Image.BICUBIC)))


def test_coefficient_tables():
    b, k = resize_ref.coefficients(640, 224)
    assert k.shape == (224, 13) and b.shape == (224, 2)           # support 2 * 640/224 = 5.71 -> ksize 13
    assert (k.sum(axis=1) > (1 << 22) - 16).all() and (k.sum(axis=1) < (1 << 22) + 16).all()
    b1, k1 = resize_ref.coefficients(224, 224)                     # identity geometry: a single unit tap
    assert all(k1[i, : b1[i, 1]].max() == 1 << 22 for i in range(224))
