"""Where the driver's 20-step window goes: host time of every `UpdatePipeline.submit` of a 20-update burst that starts from a synchronised
device (four slots, inputs_ready), the wall time of the burst, and the host time of the C call alone.

  python tools/fill_probe.py
"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import vitvs_amd
from vitvs_amd import _lib, config, synth, weights
from vitvs_amd.pipeline import UpdatePipeline
dev = torch.device("cuda", 0)
cfg = config.baseline_config("vitb16_224")
params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
sd = weights.synthetic_state_dict(cfg, 0)
des, cur = synth.frame_pair(cfg.img_size, synth.RIG8_FRAME_SEEDS[0])
I_des = torch.from_numpy(des[None]).to(dev); I_cur = torch.from_numpy(cur[None]).to(dev)
Z = torch.from_numpy(synth.depth_pattern()[None]).to(dev)
K = torch.tensor([params.intrinsics()], dtype=torch.float64, device=dev)
gen = torch.Generator().manual_seed(121)
orders = torch.stack([torch.randperm(cfg.tokens, generator=gen) for _ in range(32)]).to(torch.int32).to(dev)[:, None]
pipe = UpdatePipeline(cfg, params, sd, precision="bf16", depth=4, device=dev)
for i in range(16): pipe.submit(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 32], None, False, inputs_ready=True)
pipe.synchronize()
K_STEPS = 20
tot, subs = [], []
for rep in range(30):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); ts = []
    for i in range(K_STEPS):
        a = time.perf_counter()
        pipe.submit(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 32], None, False, inputs_ready=True)
        ts.append(time.perf_counter() - a)
    host_done = time.perf_counter() - t0
    pipe.synchronize()
    tot.append((time.perf_counter() - t0, host_done)); subs.append(ts)
tot = np.array(tot); subs = np.array(subs)
print("20 updates: wall median %.3f ms (%.0f updates/s); host finished submitting after %.3f ms" % (np.median(tot[:, 0]) * 1e3, K_STEPS / np.median(tot[:, 0]), np.median(tot[:, 1]) * 1e3))
print("host time per submit, by position (us):", np.round(np.median(subs, axis=0) * 1e6, 1).tolist())
# the C call alone
eng = pipe.engines[0]
import ctypes as C
v = pipe.v[0]; st = pipe.status[0]
with torch.cuda.stream(pipe.streams[0]):
    t = []
    for i in range(50):
        a = time.perf_counter()
        eng.compute_velocity_dev(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 32], None, False, v, st, 0)
        t.append(time.perf_counter() - a)
        pipe.streams[0].synchronize()
print("Engine.compute_velocity_dev (graph replay) host time, idle queue: median %.1f us" % (np.median(t) * 1e6))
