"""Print the per-kernel averages of a bench.py JSON line (stdin): class, launches/step, avg us."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms/step")
for k, v in sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1].get("step_share_us", 0)):
    print(f"{k:48s} x{v['launches_per_step']:5.1f}  {v['avg_us']:7.2f} us  share {v.get('step_share_us', 0):7.1f} us")
