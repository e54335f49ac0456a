"""Host logic of vit-vs_amd/pipeline.py without a GPU: slot round-robin, the ticket window, who uploads and who borrows the
weights, option calls, close order — against stand-ins for Engine and the torch stream objects (the device side is covered by
tests/test_gpu_pipeline.py)."""
import types

import pytest

import vitvs_amd  # noqa: F401
from vitvs_amd import pipeline
from vitvs_amd.engine import VitvsError


class _FakeEngine:
    log = []

    def __init__(self, cfg, params, precision="bf16", max_pairs=1, max_rows=None, device=None):
        self.device = "cuda:0"
        self.idx = len([e for e in _FakeEngine.log if e[0] == "create"])
        _FakeEngine.log.append(("create", self.idx, precision, max_pairs))

    def load_state_dict(self, sd):
        _FakeEngine.log.append(("upload", self.idx))
        return self

    def share_weights(self, owner):
        _FakeEngine.log.append(("borrow", self.idx, owner.idx))
        return self

    def set_option(self, name, value):
        _FakeEngine.log.append(("option", self.idx, name, value))
        return self

    def compute_velocity_dev(self, *a):
        _FakeEngine.log.append(("update", self.idx, a[-3] is not None, a[-1]))

    def set_goal(self, des):
        _FakeEngine.log.append(("goal", self.idx))

    def close(self):
        _FakeEngine.log.append(("close", self.idx))


class _FakeStream:
    made = []

    def __init__(self, device=None, priority=0):
        self.priority = priority
        _FakeStream.made.append(self)

    def wait_stream(self, other):
        self.waits = getattr(self, "waits", 0) + 1

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _FakeEvent:
    def record(self, st=None):
        self.recorded = True

    def synchronize(self):
        pass


@pytest.fixture
def fake(monkeypatch):
    _FakeEngine.log = []
    _FakeStream.made = []
    monkeypatch.setattr(pipeline, "Engine", _FakeEngine)
    cuda = types.SimpleNamespace(Stream=_FakeStream, Event=_FakeEvent, current_stream=lambda dev=None: _FakeStream(),
                                 stream=lambda st: st)
    zeros = lambda shape, dtype=None, device=None: types.SimpleNamespace(shape=shape, clone=lambda: "copy")  # noqa: E731
    monkeypatch.setattr(pipeline, "torch", types.SimpleNamespace(cuda=cuda, zeros=zeros, float64="f64", int32="i32",
                                                                 Tensor=object))
    return _FakeEngine


def test_one_upload_the_rest_borrow_and_every_slot_gets_its_options(fake):
    p = pipeline.UpdatePipeline("cfg", "params", {"w": 1}, depth=3)
    log = fake.log
    assert [e for e in log if e[0] == "upload"] == [("upload", 0)]
    assert [e for e in log if e[0] == "borrow"] == [("borrow", 1, 0), ("borrow", 2, 0)]
    assert [e[1:] for e in log if e[0] == "option"] == [(i, n, v) for i in range(3) for n, v in (("graph_replay", 1), ("in_flight", 3))]
    assert all(s.priority == -1 for s in _FakeStream.made[:3])      # a hardware queue each: the high-priority pool
    p.close()
    assert [e[1] for e in log if e[0] == "close"] == [2, 1, 0]        # the owner of the weights goes last


def test_copies_instead_of_borrowing_when_asked(fake):
    pipeline.UpdatePipeline("cfg", "params", {"w": 1}, depth=2, share_weights=False, plan_hint=False, graph_replay=False)
    assert [e for e in fake.log if e[0] == "upload"] == [("upload", 0), ("upload", 1)]
    assert not [e for e in fake.log if e[0] == "borrow"]
    assert [e[2:] for e in fake.log if e[0] == "option"] == [("graph_replay", 0)] * 2


def test_round_robin_and_the_ticket_window(fake):
    p = pipeline.UpdatePipeline("cfg", "params", {}, depth=3)
    with pytest.raises(VitvsError):
        p.slot(0)
    tickets = [p.submit("cur", "des", "Z", "K", 1, "order") for _ in range(7)]
    assert tickets == list(range(7))
    assert [e[1] for e in fake.log if e[0] == "update"] == [0, 1, 2, 0, 1, 2, 0]
    for t in (4, 5, 6):                                  # the last `depth` tickets are readable
        v, s, st = p.slot(t)
        assert st is p.streams[t % 3] and v is p.v[t % 3]
        assert p.result(t) == ("copy", "copy")
    for t in (0, 3, 7):                                  # overtaken, or not submitted yet
        with pytest.raises(VitvsError):
            p.result(t)
    p.set_goal("des")
    assert [e[1] for e in fake.log if e[0] == "goal"] == [0, 1, 2]
    p.join()
    p.synchronize()


def test_depth_must_be_positive(fake):
    with pytest.raises(VitvsError):
        pipeline.UpdatePipeline("cfg", "params", {}, depth=0)


def test_inputs_ready_skips_the_event_on_the_callers_stream(fake):
    """`submit(..., inputs_ready=True)`: the slot does not wait for (= records no event on) the caller's stream — on the GPU that event
    is traffic on one more hardware queue beside the slots' own (bench.py's timed region passes it; the default keeps the wait)."""
    p = pipeline.UpdatePipeline("cfg", "params", {}, depth=2)
    p.submit("cur", "des", "Z", "K")
    p.submit("cur", "des", "Z", "K", inputs_ready=True)
    p.submit("cur", "des", "Z", "K")
    slot_streams = p.streams
    assert getattr(slot_streams[0], "waits", 0) == 2 and getattr(slot_streams[1], "waits", 0) == 0


def test_set_active_uses_the_first_slots_and_restarts_the_tickets(fake):
    p = pipeline.UpdatePipeline("cfg", "params", {}, depth=4)
    assert [p.submit("c", "d", "Z", "K") for _ in range(5)] == [0, 1, 2, 3, 4]
    p.set_active(3)
    assert p.active == 3 and p.submitted == 0
    fake.log.clear()
    assert [p.submit("c", "d", "Z", "K") for _ in range(4)] == [0, 1, 2, 3]
    assert [e[1] for e in fake.log if e[0] == "update"] == [0, 1, 2, 0]          # engines 0..2 only, round-robin
    assert p.slot(3)[2] is p.streams[0]
    with pytest.raises(VitvsError):
        p.slot(0)                                                                  # rewritten by ticket 3
    with pytest.raises(VitvsError):
        p.set_active(5)
    p.set_active(4)
    assert [p.submit("c", "d", "Z", "K") for _ in range(4)] == [0, 1, 2, 3]
    assert p.slot(3)[2] is p.streams[3]
