"""Several independent servo updates in flight on one GPU.

One update at one frame pair is a chain of 86 dependent launches that leaves most of the chip idle most of the time (each
launch pays the launch-to-launch floor and its own ramp).  Updates that do not depend on each other — several cameras or
control loops sharing the GPU, or a frame stream handled as a pipeline — overlap when they are enqueued through different
handles on different HIP streams (measured on MI355X, ViT-B/16 224², bf16: 2220 updates/s on one stream, 2830 / 3110 with
2 / 3 in flight, 3390 with the 4-wave GEMM plan the ``in_flight`` hint selects; profiles/r03_notes.md section 5; round 5, the
driver's 20-step form of bench.py: 3340 / 3870 / 4020 / 3260 updates/s at depth 2 / 3 / 4 / 5 — four is the number of hardware
queues HIP gives a priority class, a fifth stream shares one; profiles/r05_driver_form_depth_sweep.txt).

Depth 4 is the best only while NOTHING else of the process keeps a queue busy: the chip runs four hardware queues side by side and
time-slices a fifth.  A caller that reads every result through torch's default stream (``result()``'s clones, a ``.cpu()``), copies
its inputs there, or runs an RCCL collective per update on the communicator's stream has that fifth queue — measured with a host read
per update, 8000 updates: 3960-4107 updates/s at depth 3 against 3652-3663 at depth 4 (profiles/r05_notes.md section 10).  Hence the
default of 3 here; bench.py passes 4 for `value`, whose timed region touches no other stream (``submit(inputs_ready=True)``; the per-update
all-gather of N > 1 is a synchronous collective, which torch >= 2.8 launches on the current = the slot's stream).

``UpdatePipeline`` is that arrangement: ``depth`` handles (own workspaces, one call in flight per handle, include/vitvs.h;
the weights are uploaded once and borrowed by the others, vitvs_share_weights) on ``depth`` streams, filled round-robin from ONE host thread, which hipGraph replay
makes cheap enough (~50 us of host time per update).  Every update is the same computation as ``Engine.compute_velocity_dev``
— the reference's ``detect_features`` + ``ibvs`` up to the raw twist (vitvs_v2.py:464-523, 588-622) — and its results
are bit-identical to the one-stream call's with the same plan (tests/test_gpu_pipeline.py).

PyTorch is plumbing here (streams, events, device buffers); all arithmetic runs in libvitvs_hip.so.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _lib
from .config import ServoParams, ViTConfig
from .engine import Engine, VitvsError


class UpdatePipeline:
    """``depth`` updates in flight.  ``submit`` enqueues one update and returns a ticket; ``result(ticket)`` waits for that
    update alone.  ``precision`` defaults to fp16: bf16 runs the same kernels at the same speed, but on weights with trained-like
    statistics its arg-max correspondences follow the fp32 reference's far less closely.  A slot's output buffers are reused every ``depth`` submissions: read (or ``result``) a ticket before
    submitting ``depth`` more."""

    def __init__(self, cfg: ViTConfig, params: ServoParams, state_dict, *, precision: str = "fp16", depth: int = 3,
                 max_pairs: int = 1, max_rows: Optional[int] = None, device=None, graph_replay: bool = True,
                 plan_hint: bool = True, stream_priority: int = -1, share_weights: bool = True,
                 stage_inputs: bool = False, streams: Optional[List[torch.cuda.Stream]] = None):
        if depth < 1:
            raise VitvsError("depth must be >= 1")
        self.depth = int(depth)
        self.active = self.depth                 # slots in use (set_active): submissions go round-robin over the first `active`
        self.engines: List[Engine] = []
        for _ in range(self.depth):
            e = Engine(cfg, params, precision=precision, max_pairs=max_pairs, max_rows=max_rows, device=device)
            if self.engines and share_weights:
                e.share_weights(self.engines[0])           # one resident copy of the weights for every queue
            else:
                e.load_state_dict(state_dict)
            e.set_option("graph_replay", int(graph_replay))
            if plan_hint:
                e.set_option("in_flight", self.depth)
            self.engines.append(e)
        self.device = self.engines[0].device
        # HIP hands out hardware queues per priority class (GPU_MAX_HW_QUEUES = 4 of each by default) and lets further
        # streams of a class SHARE them: streams of the default class compete with the caller's, torch's and RCCL's streams for
        # the same four queues, and two slots that land on one queue run one after the other (measured: depth 3 at 2500
        # instead of 3300 updates/s, depending on what else the process had created).  High-priority streams draw from a pool
        # of their own, so up to four slots always get a queue each.
        # (`streams`: run on another pipeline's streams instead of creating more — a second pipeline of the same process, e.g. another
        # precision measured beside the first, would otherwise put 2 x depth high-priority streams on the class's four queues.)
        if streams is not None and len(streams) != self.depth:
            raise VitvsError("streams: one per slot")
        # (More than four queues do not help: measured in round 5, 300-step bench — slots 5+ on default-class streams, i.e. on other
        # hardware queues: 1463 / 1763 / 1943 updates/s at depth 5 / 6 / 8 against 4462 at depth 4; GPU_MAX_HW_QUEUES=8 with every slot
        # high-priority: 1573 / 1833 / 2373.  With more than four hardware queues active the scheduler time-slices them; five streams
        # SHARING four queues lose less, 3429-3502.  profiles/r05_notes.md section 10.)
        self.streams = list(streams) if streams is not None else \
            [torch.cuda.Stream(device=self.device, priority=stream_priority) for _ in range(self.depth)]
        self.done = [torch.cuda.Event() for _ in range(self.depth)]
        n = max_pairs
        self.v = [torch.zeros((n, 6), dtype=torch.float64, device=self.device) for _ in range(self.depth)]
        self.status = [torch.zeros(n, dtype=torch.int32, device=self.device) for _ in range(self.depth)]
        self.submitted = 0
        # A replayed graph is keyed on the call's argument pointers (include/vitvs.h): a caller whose frames arrive in a new
        # buffer every update would re-capture every time.  With stage_inputs the slot owns its input buffers and `submit`
        # copies into them on the slot's stream (five small device-to-device copies), so every call of a slot replays.
        self.stage_inputs = bool(stage_inputs)
        self._staged = [dict() for _ in range(self.depth)]

    def _stable(self, k: int, name: str, t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        if t is None or not self.stage_inputs:
            return t
        buf = self._staged[k].get(name)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty_like(t)
            self._staged[k][name] = buf
        buf.copy_(t, non_blocking=True)
        t.record_stream(torch.cuda.current_stream(self.device))   # the caller may drop `t` at once: its memory is not reused
        return buf                                                # before this (the slot's) stream has copied it

    def close(self):
        for e in reversed(self.engines):                 # borrowers first, the owner of the weights last
            e.close()
        self.engines = []

    # ------------------------------------------------------------------ enqueue
    def submit(self, I_cur: torch.Tensor, I_des: Optional[torch.Tensor], Z: Optional[torch.Tensor], K: torch.Tensor,
               mode: int = _lib.SELECT_DENSE, selection: Optional[torch.Tensor] = None,
               n_selected: Optional[torch.Tensor] = None, des_shared: bool = False, num_pairs: int = 0,
               inputs_ready: bool = False) -> int:
        """Arguments as ``Engine.compute_velocity_dev`` (device tensors).  The slot's stream first waits for the caller's
        current stream, so inputs produced there are complete; nothing synchronises the host.  ``inputs_ready``: the caller
        vouches that the inputs are complete already (e.g. device-resident buffers written before a synchronisation) — no event is
        recorded on the caller's stream, which otherwise is one more hardware queue with traffic on every update (module docstring).
        Without ``stage_inputs`` the update reads the caller's tensors in place: keep them alive and unchanged until the ticket has
        completed (``result``)."""
        t = self.submitted
        k = t % self.active
        st = self.streams[k]
        if not inputs_ready:
            st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            I_cur, I_des, Z, K = (self._stable(k, n, t) for n, t in (("cur", I_cur), ("des", I_des), ("Z", Z), ("K", K)))
            selection, n_selected = self._stable(k, "sel", selection), self._stable(k, "nsel", n_selected)
            self.engines[k].compute_velocity_dev(I_cur, I_des, Z, K, mode, selection, n_selected, des_shared,
                                                 self.v[k], self.status[k], num_pairs)
            self.done[k].record(st)
        self.submitted = t + 1
        return t

    def set_active(self, n: int):
        """Use only the first ``n`` slots from now on (1 <= n <= depth): the pipeline is drained and ticket numbering starts again at 0.
        For a caller that has to find out at run time how many queues are its own — the chip runs four side by side, and whatever else
        of the process keeps a queue busy (a collective on a communicator's stream, copies on another stream) counts (module docstring);
        bench.py calibrates its multi-GPU line with it."""
        if not 1 <= int(n) <= self.depth:
            raise VitvsError("active slots: 1 .. depth")
        self.synchronize()
        self.active = int(n)
        self.submitted = 0

    def set_goal(self, I_des: torch.Tensor):
        """Cache the goal frame(s) in every handle (``Engine.set_goal``): later ``submit(..., I_des=None, ...)``."""
        for k, e in enumerate(self.engines):
            self.streams[k].wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.streams[k]):
                e.set_goal(I_des)

    # ------------------------------------------------------------------ collect
    def slot(self, ticket: int) -> Tuple[torch.Tensor, torch.Tensor, torch.cuda.Stream]:
        """Output buffers (v_c [n, 6] float64, status [n] int32) and stream of ``ticket`` without waiting."""
        if not (self.submitted - self.active <= ticket < self.submitted):
            raise VitvsError(f"ticket {ticket} is not in flight (submitted {self.submitted}, depth {self.active})")
        k = ticket % self.active
        return self.v[k], self.status[k], self.streams[k]

    def result(self, ticket: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Wait (host) for ``ticket`` and return copies of its ``v_c`` and ``status``."""
        v, s, _ = self.slot(ticket)
        self.done[ticket % self.active].synchronize()
        return v.clone(), s.clone()

    def join(self, stream: Optional[torch.cuda.Stream] = None):
        """Make ``stream`` (default: the current one) wait for everything submitted so far (device-side)."""
        stream = stream or torch.cuda.current_stream(self.device)
        for k in range(min(self.active, self.submitted)):
            stream.wait_event(self.done[k])

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
