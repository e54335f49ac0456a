mkdir -p gpurun_out/r03i
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" -p no:cacheprovider > gpurun_out/r03i/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r03i/pytest.log
tail -3 gpurun_out/r03i/pytest.log
for rep in 1 2; do
echo "== product (dot2) $rep" >> gpurun_out/r03i/attn.txt; timeout -k 10 200 tools/big_ops attn >> gpurun_out/r03i/attn.txt 2>&1
echo "== adds $rep" >> gpurun_out/r03i/attn.txt; LD_LIBRARY_PATH=vit-vs_amd/variants/adds timeout -k 10 200 tools/big_ops attn >> gpurun_out/r03i/attn.txt 2>&1
done
grep -E "==|^attention" gpurun_out/r03i/attn.txt | grep -E "==|3137|1370|8 x 785"
