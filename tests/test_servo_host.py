"""Host-side mirror of the reference interface (vit-vs_amd/servo.py) on CPU: the selection glue must reproduce
the reference's draw (golden fixtures) when the GPU stage is replaced by the oracle's arg-max tables."""
import numpy as np
import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import servo
from oracle import servo_ref as sr
from conftest import golden_case, load_golden


class _OracleBackedEngine:
    """Stands in for Engine.correspond only (similarities + arg-max from the CPU oracle)."""

    def correspond(self, d1, d2):
        sim = sr.cosine_matrix(d1, d2)
        sim_1, nn_1, _, nn_2 = sr.nearest_neighbours(sim)
        return nn_1.int(), nn_2.int(), sim_1


@pytest.mark.parametrize("name", ["partial", "short", "tiny", "grid14", "all_mutual", "same_image"])
def test_find_correspondences_batch_reproduces_reference_draw(name):
    case = golden_case(load_golden("corr_cases.npz"), name)
    d1 = torch.from_numpy(case["desc1"])[None, None]
    d2 = torch.from_numpy(case["desc2"])[None, None]
    torch.manual_seed(121)
    p1, p2, sim = servo.find_correspondences_batch(_OracleBackedEngine(), d1, d2, num_pairs=int(case["num_pairs"]))
    if int(case["status"]) == 1:
        assert p1 is None and p2 is None and sim is None
        return
    assert np.array_equal(p1.numpy(), case["points1"]) and np.array_equal(p2.numpy(), case["points2"])
    np.testing.assert_array_equal(sim.reshape(-1).numpy(), case["sim_selected"])


def test_candidate_order_is_mutual_nn_set():
    case = golden_case(load_golden("corr_cases.npz"), "grid14")
    nn1, nn2 = torch.from_numpy(case["nn_1"]).long(), torch.from_numpy(case["nn_2"]).long()
    cand = servo._candidate_order(nn1, nn2, 14)
    mutual = torch.nonzero(nn2[nn1] == torch.arange(196)).flatten()
    assert sorted(cand.tolist()) == mutual.tolist()


def test_ema_and_twist_match_reference_fixture():
    case = golden_case(load_golden("corr_cases.npz"), "partial")
    state = [None] * 6
    np.testing.assert_array_equal(servo.ema_update(state, case["v_c"], 0.8), case["ema_first"])
    np.testing.assert_allclose(servo.ema_update(state, 0.5 * case["v_c"], 0.8), case["ema_second"], rtol=1e-15)
    lin, ang = servo.twist_from_velocity([0.1, -0.2, 3.0, 0.4, -0.5, 0.6], 1.0)
    assert lin == (1.0, -0.1, 0.2) and ang == (0.6, -0.4, 0.5)
    assert (lin, ang) == sr.twist_remap([0.1, -0.2, 3.0, 0.4, -0.5, 0.6], 1.0)
