mkdir -p gpurun_out/r03z
export HIP_FORCE_DEV_KERNARG=1
python __graft_entry__.py --smoke > gpurun_out/r03z/smoke.log 2>&1; tail -2 gpurun_out/r03z/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03z/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03z/pytest.log; tail -3 gpurun_out/r03z/pytest.log
( time python bench.py > gpurun_out/r03z/bench.json 2> gpurun_out/r03z/bench.err ) 2> gpurun_out/r03z/bench_time.txt; tail -3 gpurun_out/r03z/bench_time.txt; python -c "
import json
d=json.loads(open('gpurun_out/r03z/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('recomputed_from_profile'), d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['thread_counts_not_timed'], d['cpu_baseline']['cgroup_cpus'])
"
LD_LIBRARY_PATH=vit-vs_amd/variants/probe tools/big_ops attn > gpurun_out/r03z/attention_probe.txt 2>&1; head -8 gpurun_out/r03z/attention_probe.txt | cut -c1-330
