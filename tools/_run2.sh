mkdir -p gpurun_out/r03b
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_loop.py -m gpu -q -s -k "attention or loop" -p no:cacheprovider > gpurun_out/r03b/pytest_attn.log 2>&1; echo "rc $?" >> gpurun_out/r03b/pytest_attn.log
tail -5 gpurun_out/r03b/pytest_attn.log
timeout -k 10 500 python -m pytest tests/test_gpu_path.py -m gpu -q -s -k "trained_like or fp16_large or 16bit_many or graph or many_tokens or 3136" -p no:cacheprovider > gpurun_out/r03b/pytest_path.log 2>&1; echo "rc $?" >> gpurun_out/r03b/pytest_path.log
tail -5 gpurun_out/r03b/pytest_path.log
echo "== auto" > gpurun_out/r03b/attn.txt; timeout -k 10 200 tools/big_ops attn >> gpurun_out/r03b/attn.txt 2>&1
for g in g0 g256 g512 g768; do echo "== $g" >> gpurun_out/r03b/attn.txt; LD_LIBRARY_PATH=vit-vs_amd/variants/$g timeout -k 10 200 tools/big_ops attn >> gpurun_out/r03b/attn.txt 2>&1; done
grep -E "==|attention" gpurun_out/r03b/attn.txt
