"""Do independent updates overlap when they are queued on two streams?  (tools only, never product)

One update at one frame pair is 86 dependent launches of ~5 us each; consecutive updates of a throughput run do not
depend on each other.  This script times the same number of updates
  A  one handle, one stream, eager launches            (what bench.py's `value` is)
  B  one handle, one stream, hipGraph replay           (VITVS_GRAPH=1)
  C  two handles on two streams, graph replay, one host thread alternating between them
  D  two handles on two streams, eager launches from two host threads
  E  as C with three handles / streams
and prints updates/s for each.  Round 1 saw the two FRAMES of one update, put on two queues, alternate instead of overlap
(profiles/r01_notes.md); this is the same question for whole updates.
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vitvs_amd  # noqa: F401
from vitvs_amd import _lib, config, synth, weights
from vitvs_amd.engine import Engine


def make(cfg, params, sd, graph, prec="bf16", pairs=1):
    os.environ["VITVS_GRAPH"] = "1" if graph else "0"
    e = Engine(cfg, params, precision=prec, max_pairs=pairs).load_state_dict(sd)
    if os.environ.get("TOOL_IN_FLIGHT"):
        e.set_option("in_flight", int(os.environ["TOOL_IN_FLIGHT"]))
    return e


def main():
    dev = torch.device("cuda", 0)
    name = sys.argv[1] if len(sys.argv) > 1 else "vitb16_224"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    cfg = config.baseline_config(name)
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 0)
    des, cur = synth.frame_pair(cfg.img_size, synth.RIG8_FRAME_SEEDS[0])
    depth = synth.depth_pattern()
    I_des = torch.from_numpy(des[None]).to(dev)
    I_cur = torch.from_numpy(cur[None]).to(dev)
    Z = torch.from_numpy(depth[None]).to(dev)
    K = torch.tensor([params.intrinsics()], dtype=torch.float64, device=dev)
    gen = torch.Generator().manual_seed(121)
    orders = torch.stack([torch.randperm(cfg.tokens, generator=gen) for _ in range(64)]).to(torch.int32).to(dev)[:, None]

    def slot():
        return torch.zeros((1, 6), dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)

    def run_one_thread(engines, streams, n):
        outs = [slot() for _ in engines]
        def go(count):
            for i in range(count):
                k = i % len(engines)
                with torch.cuda.stream(streams[k]):
                    engines[k].compute_velocity_dev(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 64], None, False,
                                                    outs[k][0], outs[k][1])
        go(40)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        go(n)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        return n / dt, t_host / n * 1e6, [o[0].cpu().numpy().copy() for o in outs]

    def run_threads(engines, streams, n):
        outs = [slot() for _ in engines]
        def worker(k, count):
            with torch.cuda.stream(streams[k]):
                for i in range(count):
                    engines[k].compute_velocity_dev(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 64], None, False,
                                                    outs[k][0], outs[k][1])
        def go(count):
            ts = [threading.Thread(target=worker, args=(k, count // len(engines))) for k in range(len(engines))]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        go(40)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        go(n)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        return n / dt, 0.0, [o[0].cpu().numpy().copy() for o in outs]

    s = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(6)]   # a hardware queue each (vit-vs_amd/pipeline.py)
    res = {}
    e_eager = [make(cfg, params, sd, False) for _ in range(4)]
    e_graph = [make(cfg, params, sd, True) for _ in range(6)]
    if len(sys.argv) > 3:                     # quick form: eager on one stream, then graph replay on the listed stream counts
        for rep in range(2):
            line = [f"eager x1 {run_one_thread(e_eager[:1], s[:1], steps)[0]:7.1f}"]
            for k in [int(t) for t in sys.argv[3].split(",")]:
                line.append(f"graph x{k} {run_one_thread(e_graph[:k], s[:k], steps - steps % k)[0]:7.1f}")
            print(f"rep {rep}  " + "   ".join(line), flush=True)
        return
    for rep in range(2):
        res["A eager 1 stream"] = run_one_thread(e_eager[:1], s[:1], steps)
        res["B graph 1 stream"] = run_one_thread(e_graph[:1], s[:1], steps)
        res["C graph 2 streams, 1 host thread"] = run_one_thread(e_graph[:2], s[:2], steps)
        res["E graph 3 streams, 1 host thread"] = run_one_thread(e_graph[:3], s[:3], steps)
        res["H graph 4 streams, 1 host thread"] = run_one_thread(e_graph[:4], s[:4], steps)
        res["I graph 6 streams, 1 host thread"] = run_one_thread(e_graph[:6], s[:6], steps - steps % 6)
        res["J eager 4 streams, 4 host threads"] = run_threads(e_eager[:4], s[:4], steps)
        res["D eager 2 streams, 2 host threads"] = run_threads(e_eager[:2], s[:2], steps)
        res["F graph 2 streams, 2 host threads"] = run_threads(e_graph[:2], s[:2], steps)
        res["G eager 3 streams, 3 host threads"] = run_threads(e_eager[:3], s[:3], steps - steps % 3)
        ref = res["A eager 1 stream"][2][0]
        for k, (ups, host_us, vs) in res.items():
            same = all(np.array_equal(v, vs[0]) for v in vs)
            print(f"rep {rep}  {k:40s} {ups:8.1f} updates/s   host {host_us:6.1f} us/update   outputs identical across handles: {same}",
                  flush=True)


if __name__ == "__main__":
    main()
