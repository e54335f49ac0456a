#!/bin/bash
# One gpurun call that refreshes everything under profiles/ for a round (run from the repo root on the GPU box):
#   bench lines (bf16 with CPU baseline, fp32), rocprofv3 kernel trace + stats, three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES), micro-benchmarks.
# usage: tools/measure_round.sh r01
R="${1:-r01}"; O=gpurun_out/$R; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
make -C tools > /dev/null 2>&1
python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err || exit 1
python bench.py --precision fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err || exit 1
for b in 2 4 8; do python bench.py --pairs $b --no-cpu-baseline --steps 100 > $O/bench_bf16_pairs$b.json 2>/dev/null || exit 1; done
python bench.py --precision fp16 --no-cpu-baseline > $O/bench_fp16.json 2>/dev/null || exit 1
# the other BASELINE.json configurations (parity-test cases; not the headline)
for c in vitb8_448 vitl14_518 vits14_308 vits16_224; do python bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-plain-chain > $O/bench_bf16_$c.json 2>/dev/null || exit 1; done
python bench.py --config vitb8_448 --selection dense --steps 50 --warmup 5 --no-cpu-baseline --no-plain-chain > $O/bench_bf16_vitb8_448_dense.json 2>/dev/null || exit 1
python bench.py --config vitl14_518 --precision fp16 --steps 50 --warmup 5 --no-cpu-baseline --no-plain-chain > $O/bench_fp16_vitl14_518.json 2>/dev/null || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-plain-chain > $GRAFT_REPO_ROOT/$O/bench_bf16_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof_trace.err ) || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_fetch.err ) || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_write.err ) || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_mfma.err ) || exit 1
tools/launch_floor > $O/launch_floor.txt 2>&1
tools/op_chain > $O/op_chain_bf16.txt 2>&1
tools/op_chain 32 > $O/op_chain_fp32.txt 2>&1
tools/intake_bench > $O/intake_bench.txt 2>&1
find $O -name "*.csv" | head -20
# keep only the small summaries in the merged output
python tools/trace_summary.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/trace_summary_bf16.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16.csv
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) > $O/pmc_traffic.json
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_mfma
tail -c 600 $O/bench_bf16.json
