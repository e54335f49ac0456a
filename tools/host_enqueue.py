"""Host time to enqueue one update (86 launches through one C-ABI call) vs the GPU time of the update."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vitvs_amd
from vitvs_amd import _lib, config, synth, weights
from vitvs_amd.engine import Engine

cfg = config.baseline_config("vitb16_224")
params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
eng = Engine(cfg, params, precision="bf16", max_pairs=1).load_state_dict(weights.synthetic_state_dict(cfg, 0))
dev = torch.device("cuda", 0)
des, cur = synth.frame_pair(cfg.img_size, synth.ACCEPTED_FRAME_SEEDS["vitb16_224"])
I_des, I_cur = torch.from_numpy(des[None]).to(dev), torch.from_numpy(cur[None]).to(dev)
Z = torch.from_numpy(synth.depth_pattern()[None]).to(dev)
K = torch.tensor([params.intrinsics()], dtype=torch.float64, device=dev)
order = torch.randperm(cfg.tokens).to(torch.int32).to(dev)[None]
v = torch.zeros((1, 6), dtype=torch.float64, device=dev); status = torch.zeros(1, dtype=torch.int32, device=dev)
def step(): eng.compute_velocity_dev(I_cur, I_des, Z, K, _lib.SELECT_ORDER, order, None, False, v, status)
for _ in range(50): step()
torch.cuda.synchronize()
# host-only cost: enqueue into an idle queue and measure until the call returns (GPU runs behind)
ts = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
print(f"host enqueue of one update into an idle queue: median {np.median(ts)*1e6:.1f} us, min {min(ts)*1e6:.1f} us")
t0 = time.perf_counter()
for _ in range(300): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"300 updates: host loop returned after {(t1-t0)/300*1e6:.1f} us/update, GPU done after {(t2-t0)/300*1e6:.1f} us/update")
