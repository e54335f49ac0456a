"""bench.py's `roofline.overlapped` measurement (captured proj / fc2 chains of the dominant GEMM symbol, one chain per queue) for 1, 2, 3
and 4 queues, with N high-priority streams already created in the process (the pipeline's own) — the record behind
profiles/r05_overlap_queues.txt: the chain's per-launch time is not monotonic in the queue count (3 queues are slower than 2, 4 much
faster), whatever streams exist beside it.

  python tools/overlap_probe.py [pre-existing high-priority streams]
"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import bench
import vitvs_amd
from vitvs_amd import _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
held = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(extra)]   # pre-existing high-priority streams, like the pipeline's
for q in (1, 2, 3, 4):
    lib.vitvs_op_plan_in_flight(max(q, 1))
    s_proj = bench.split_k(394, 768, 768, 64, q, True)
    s_fc2 = bench.split_k(394, 768, 3072, 64, q, True)
    r = bench.overlapped_chain_us("bf16", 394, 768, 3072, s_proj, s_fc2, dev, queues=q)
    print(f"pre-existing high-priority streams {extra}, queues {q}: splits {s_proj}/{s_fc2}", json.dumps(r))
