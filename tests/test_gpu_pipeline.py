"""Several updates in flight (vit-vs_amd/pipeline.py, vitvs_set_option): the pipelined updates are the SAME computation
as the one-stream call — bit-identical to it under the same tile plan, and within the parity bars of tests/test_gpu_path.py
against the CPU oracle under the 4-wave plan the ``in_flight`` hint selects."""
import numpy as np
import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import _lib, config, synth, weights
from vitvs_amd.engine import Engine, VitvsError
from vitvs_amd.pipeline import UpdatePipeline
from conftest import golden_case, load_golden

pytestmark = pytest.mark.gpu


def _inputs(cfg, params, seeds, dev):
    pairs = [synth.frame_pair(cfg.img_size, s) for s in seeds]
    depth = synth.depth_pattern()
    des = [torch.from_numpy(p[0][None]).to(dev) for p in pairs]
    cur = [torch.from_numpy(p[1][None]).to(dev) for p in pairs]
    Z = torch.from_numpy(depth[None]).to(dev)
    K = torch.tensor([params.intrinsics()], dtype=torch.float64, device=dev)
    return pairs, depth, des, cur, Z, K


def _orders(cfg, n, dev, seed=121):
    gen = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randperm(cfg.tokens, generator=gen) for _ in range(n)]).to(torch.int32).to(dev)[:, None]


@pytest.mark.parametrize("precision", ["bf16", "fp32", "f16x2"])
@pytest.mark.parametrize("hint,depth", [(False, 3), (True, 3), (True, 4)])
def test_pipelined_updates_equal_the_one_stream_call_bit_for_bit(precision, hint, depth):
    """11 updates over 4 different frame pairs and 11 visiting orders through a pipeline of depth 3 (rounds 3-4's bench protocol) or 4
    (bench.py's default since round 5) against the same updates, one at a time, through a single handle with the same plan."""
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config("vits16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 3)
    seeds = synth.RIG8_FRAME_SEEDS[:4]
    _, _, des, cur, Z, K = _inputs(cfg, params, seeds, dev)
    n = 11
    orders = _orders(cfg, n, dev)

    eng = Engine(cfg, params, precision=precision, max_pairs=1).load_state_dict(sd)
    if hint:
        eng.set_option("in_flight", depth)
    want = []
    for i in range(n):
        v, s = eng.compute_velocity_dev(cur[i % 4], des[i % 4], Z, K, _lib.SELECT_ORDER, orders[i])
        want.append((v.cpu().numpy().copy(), s.cpu().numpy().copy()))
    assert all(int(s[0]) == _lib.STATUS_OK for _, s in want)
    assert len({w[0].tobytes() for w in want}) > 4        # the updates really differ (pairs and orders)

    pipe = UpdatePipeline(cfg, params, sd, precision=precision, depth=depth, plan_hint=hint)
    got = {}
    tickets = []
    for i in range(n):
        tickets.append(pipe.submit(cur[i % 4], des[i % 4], Z, K, _lib.SELECT_ORDER, orders[i]))
        if len(tickets) == depth:                               # never more than `depth` unread results
            t = tickets.pop(0)
            got[t] = pipe.result(t)
    for t in tickets:
        got[t] = pipe.result(t)
    for i in range(n):
        v, s = got[i]
        assert np.array_equal(v.cpu().numpy(), want[i][0]), f"update {i}"
        assert np.array_equal(s.cpu().numpy(), want[i][1])
    # a second pass replays the captured graphs: still the same bits
    t2 = [pipe.submit(cur[i % 4], des[i % 4], Z, K, _lib.SELECT_ORDER, orders[i]) for i in range(depth)]
    for i, t in enumerate(t2):
        assert np.array_equal(pipe.result(t)[0].cpu().numpy(), want[i][0])
    pipe.close()
    eng.close()


@pytest.mark.parametrize("key", ["vits16_224", "vitb16_224"])
def test_in_flight_plan_meets_the_reference_fixture_in_fp32(key):
    """The 4-wave GEMM plan changes only the summation order inside a GEMM's k-loop: against the reference-generated fixture the
    fp32 mode keeps bit-exact arg-max tables and pixel features and v_c <= 1e-9 (north_star: 1e-4), exactly as
    tests/test_gpu_path.py::test_compute_velocity_fp32_matches_reference asserts for the default plan."""
    blob = load_golden(f"e2e_{key}.npz")
    case = golden_case(blob, "plain")
    cfg = config.baseline_config(key)
    sd = weights.synthetic_state_dict(cfg, int(blob["weight_seed"]))
    des, cur = synth.frame_pair(cfg.img_size, int(blob["frame_seed"]))
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    eng = Engine(cfg, params, precision="fp32", max_pairs=1).load_state_dict(sd).set_option("in_flight", 3)
    sel = (case["points1"][:, 0] * cfg.grid + case["points1"][:, 1]).astype(np.int32)
    v, st = eng.compute_velocity(cur, des, synth.depth_pattern(), params.intrinsics(), mode=_lib.SELECT_EXPLICIT, selection=[sel])
    det = eng.last_details(1)
    assert int(st[0]) == 0
    assert np.array_equal(det["nn_1"][0], case["nn_1"]) and np.array_equal(det["nn_2"][0], case["nn_2"])
    np.testing.assert_allclose(det["sim_1"][0], case["sim_1"], rtol=0, atol=2e-5)
    k = case["s_uv"].shape[0]
    assert np.array_equal(det["s_uv"][0, :k, 2:4], case["s_uv"]) and np.array_equal(det["s_uv"][0, :k, 0:2], case["s_uv_star"])
    got, want = v.cpu().numpy()[0], case["v_c"]
    assert np.linalg.norm(got - want) <= 1e-9 * np.linalg.norm(want)
    eng.close()


def test_staged_inputs_replay_one_graph_per_slot_whatever_buffers_the_caller_brings():
    """Frames arriving in a fresh device buffer every update (a camera driver's ring): with stage_inputs the slot copies them into
    buffers of its own, every call of a slot replays the slot's one captured graph, and the results are those of the direct call."""
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config("vits16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 3)
    _, _, des, cur, Z, K = _inputs(cfg, params, synth.RIG8_FRAME_SEEDS[:3], dev)
    orders = _orders(cfg, 9, dev)
    ref = UpdatePipeline(cfg, params, sd, precision="bf16", depth=2)
    want = [ref.result(ref.submit(cur[i % 3], des[i % 3], Z, K, _lib.SELECT_ORDER, orders[i]))[0].cpu().numpy() for i in range(9)]
    ref.close()
    pipe = UpdatePipeline(cfg, params, sd, precision="bf16", depth=2, stage_inputs=True)
    for i in range(9):
        fresh = [t.clone() for t in (cur[i % 3], des[i % 3], Z, K, orders[i])]          # new addresses every update
        got = pipe.result(pipe.submit(fresh[0], fresh[1], fresh[2], fresh[3], _lib.SELECT_ORDER, fresh[4]))[0].cpu().numpy()
        assert np.array_equal(got, want[i]), f"update {i}"
    assert all(set(st) == {"cur", "des", "Z", "K", "sel"} for st in pipe._staged)
    pipe.close()


def test_many_token_updates_side_by_side_are_reproducible():
    """DINO ViT-B/8 448² (3137 tokens: the key-split long-sequence attention with its fence-free hand-off, the 256-row GEMM tiles, the
    f16 split Gram) with three updates in flight: every result equals the first pass of its (frame pair, visiting order), whichever
    slot computed it and whatever ran beside it."""
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config("vitb8_448")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 0)
    seeds = [synth.ACCEPTED_FRAME_SEEDS["vitb8_448"] + i for i in range(2)]
    _, _, des, cur, Z, K = _inputs(cfg, params, seeds, dev)
    P = 4                                                         # (pair, order) cases; 4 and the depth 3 are coprime
    orders = _orders(cfg, P, dev)
    pipe = UpdatePipeline(cfg, params, sd, precision="bf16", depth=3)
    first, tickets = {}, []
    for i in range(36):
        tickets.append((i, pipe.submit(cur[i % 2], des[i % 2], Z, K, _lib.SELECT_ORDER, orders[i % P])))
        if len(tickets) == 3:
            j, t = tickets.pop(0)
            v, st = pipe.result(t)
            assert int(st[0]) == _lib.STATUS_OK
            assert first.setdefault(j % P, v.cpu().numpy().tobytes()) == v.cpu().numpy().tobytes(), f"update {j}"
    for j, t in tickets:
        v, _ = pipe.result(t)
        assert first.setdefault(j % P, v.cpu().numpy().tobytes()) == v.cpu().numpy().tobytes(), f"update {j}"
    assert len(set(first.values())) == P
    pipe.close()


def test_tickets_options_and_cached_goal():
    dev = torch.device("cuda", 0)
    blob = load_golden("e2e_vitb16_224.npz")                    # an accepted pair: arg-max margins above the fp32 noise
    cfg = config.baseline_config("vitb16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, int(blob["weight_seed"]))
    _, _, des, cur, Z, K = _inputs(cfg, params, [int(blob["frame_seed"]), synth.RIG8_FRAME_SEEDS[1]], dev)
    orders = _orders(cfg, 4, dev)
    pipe = UpdatePipeline(cfg, params, sd, precision="fp32", depth=2)
    with pytest.raises(VitvsError):
        pipe.slot(0)                                            # nothing submitted yet
    e = pipe.engines[0]
    with pytest.raises(VitvsError, match="unknown option"):
        e.set_option("no_such_option", 1)
    with pytest.raises(VitvsError, match="in_flight"):
        e.set_option("in_flight", 0)
    with pytest.raises(VitvsError, match="graph_replay"):
        e.set_option("graph_replay", 2)
    # the slots borrow slot 0's weights (vitvs_share_weights): uploads go to the owner only, and only like shares with like
    sd_np = {k: v for k, v in sd.items()}
    with pytest.raises(VitvsError, match="borrows"):
        pipe.engines[1].load_state_dict(sd_np)
    other = Engine(cfg, params, precision="bf16", max_pairs=1)
    with pytest.raises(VitvsError, match="precision"):
        other.share_weights(pipe.engines[0])
    other.close()
    fresh = Engine(cfg, params, precision="fp32", max_pairs=1)
    with pytest.raises(VitvsError, match="owns"):
        fresh.share_weights(pipe.engines[1])                    # a borrower cannot lend
    fresh.close()
    # cached goal in every handle: I_des = None forwards only the current frame, on whichever slot the update lands
    with_goal = [pipe.result(pipe.submit(cur[0], des[0], Z, K, _lib.SELECT_ORDER, orders[i]))[0].cpu().numpy() for i in range(2)]
    pipe.set_goal(des[0])
    cached = [pipe.result(pipe.submit(cur[0], None, Z, K, _lib.SELECT_ORDER, orders[i]))[0].cpu().numpy() for i in range(2)]
    for a, b in zip(cached, with_goal):
        assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b)      # row count of the GEMMs differs: summation order only
    # a ticket that has been overtaken by `depth` later submissions is refused
    t0 = pipe.submit(cur[1], des[1], Z, K, _lib.SELECT_ORDER, orders[2])
    pipe.submit(cur[1], des[1], Z, K, _lib.SELECT_ORDER, orders[3])
    pipe.submit(cur[0], des[0], Z, K, _lib.SELECT_ORDER, orders[0])
    with pytest.raises(VitvsError):
        pipe.result(t0)
    pipe.synchronize()
    pipe.close()


def test_borrowed_weights_outlive_their_lender():
    """vitvs_share_weights shares OWNERSHIP (include/vitvs.h): the device weights are released by the last handle holding them, so
    closing the lender first — an explicit owner.close(), which Python's garbage collector does not guard against — leaves the
    borrowers working (it used to be a use-after-free), with eager launches and with a replayed graph that holds the addresses."""
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config("vits16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 3)
    _, _, des, cur, Z, K = _inputs(cfg, params, synth.RIG8_FRAME_SEEDS[:1], dev)
    order = _orders(cfg, 1, dev)[0]
    owner = Engine(cfg, params, precision="bf16", max_pairs=1).load_state_dict(sd)
    want, _ = owner.compute_velocity_dev(cur[0], des[0], Z, K, _lib.SELECT_ORDER, order)
    want = want.cpu().numpy().copy()
    a = Engine(cfg, params, precision="bf16", max_pairs=1).share_weights(owner)
    b = Engine(cfg, params, precision="bf16", max_pairs=1).share_weights(owner).set_option("graph_replay", 1)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        vb, _ = b.compute_velocity_dev(cur[0], des[0], Z, K, _lib.SELECT_ORDER, order)      # captures b's graph while the lender lives
        st.synchronize()
    owner.close()                                                  # the lender goes first
    filler = [torch.full((64, 1024, 1024), 7.0, device=dev) for _ in range(4)]               # 1 GiB of fresh allocations over whatever was freed
    torch.cuda.synchronize()
    va, _ = a.compute_velocity_dev(cur[0], des[0], Z, K, _lib.SELECT_ORDER, order)
    with torch.cuda.stream(st):
        vb2, _ = b.compute_velocity_dev(cur[0], des[0], Z, K, _lib.SELECT_ORDER, order)     # replay
        st.synchronize()
    torch.cuda.synchronize()
    assert np.array_equal(va.cpu().numpy(), want) and np.array_equal(vb.cpu().numpy(), want) and np.array_equal(vb2.cpu().numpy(), want)
    with pytest.raises(VitvsError, match="borrows"):
        a.load_state_dict(sd)                                       # still a borrower: nothing can upload to those weights any more
    del filler
    a.close()
    b.close()


def test_a_borrower_that_changes_lender_replays_over_the_new_weights():
    """A handle that borrowed from one owner and then borrows from another drops the updates it captured over the first owner's
    weights: its next call (graph replay on, same argument tuple) runs on the second owner's."""
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config("vits16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    _, _, des, cur, Z, K = _inputs(cfg, params, synth.RIG8_FRAME_SEEDS[:1], dev)
    order = _orders(cfg, 1, dev)[0]
    owners = [Engine(cfg, params, precision="bf16", max_pairs=1).load_state_dict(weights.synthetic_state_dict(cfg, s)) for s in (3, 4)]
    want = [o.compute_velocity_dev(cur[0], des[0], Z, K, _lib.SELECT_ORDER, order)[0].cpu().numpy().copy() for o in owners]
    assert not np.array_equal(want[0], want[1])
    b = Engine(cfg, params, precision="bf16", max_pairs=1).share_weights(owners[0]).set_option("graph_replay", 1)
    st = torch.cuda.Stream()
    got = []
    with torch.cuda.stream(st):
        for o in (owners[0], owners[0], owners[1], owners[1]):
            b.share_weights(o)
            v, _ = b.compute_velocity_dev(cur[0], des[0], Z, K, _lib.SELECT_ORDER, order)
            st.synchronize()
            got.append(v.cpu().numpy().copy())
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[0])
    assert np.array_equal(got[2], want[1]) and np.array_equal(got[3], want[1])
    b.close()
    for o in owners:
        o.close()
