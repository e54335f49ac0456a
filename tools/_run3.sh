mkdir -p gpurun_out/r03e
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" -p no:cacheprovider > gpurun_out/r03e/pytest_attn.log 2>&1; echo "rc $?" >> gpurun_out/r03e/pytest_attn.log
tail -4 gpurun_out/r03e/pytest_attn.log
timeout -k 10 600 python -m pytest tests/test_gpu_path.py -m gpu -q -s -k "16bit or facets or fp16_large or rig_of_8 or graph_replay" -p no:cacheprovider > gpurun_out/r03e/pytest_path.log 2>&1; echo "rc $?" >> gpurun_out/r03e/pytest_path.log
tail -4 gpurun_out/r03e/pytest_path.log
echo "== product" > gpurun_out/r03e/attn.txt; timeout -k 10 200 tools/big_ops attn >> gpurun_out/r03e/attn.txt 2>&1
echo "== fences" >> gpurun_out/r03e/attn.txt; LD_LIBRARY_PATH=vit-vs_amd/variants/fences timeout -k 10 200 tools/big_ops attn >> gpurun_out/r03e/attn.txt 2>&1
grep -E "==|^attention" gpurun_out/r03e/attn.txt
