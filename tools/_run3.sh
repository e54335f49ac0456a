mkdir -p gpurun_out/r03l
export HIP_FORCE_DEV_KERNARG=1
R=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" -p no:cacheprovider > gpurun_out/r03l/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r03l/pytest.log; tail -3 gpurun_out/r03l/pytest.log
run() { (cd $1 && python bench.py $2 --no-cpu-baseline --no-secondary --no-plain-chain 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if k in ('attention',)})"); }
for i in 1 2; do
  for cfg in "--config vitb8_448 --steps 50 --warmup 5" "--config vitl14_518 --steps 50 --warmup 5" "--pairs 8 --steps 100"; do
    echo "round $i [$cfg] r02: $(run $R/_r02 "$cfg")" >> gpurun_out/r03l/ab.txt
    echo "round $i [$cfg] new: $(run $R "$cfg")" >> gpurun_out/r03l/ab.txt
  done
done
cat gpurun_out/r03l/ab.txt
