#!/bin/bash
# One gpurun call that refreshes everything under profiles/ for a round (run from the repo root on the GPU box):
#   bench lines (bf16 with CPU baseline + fp32 secondary), the other BASELINE.json configurations, rocprofv3 kernel trace +
#   stats, three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES), micro-benchmarks.
# usage: tools/measure_round.sh r05 <git commit> [bench|bench2|prof|micro]   (one gpurun call each: a call is limited to 20 minutes)
R="${1:-r04}"; COMMIT="${2:-unknown}"; PART="${3:-all}"; O=gpurun_out/$R; mkdir -p $O
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
export HIP_FORCE_DEV_KERNARG=1
make -C tools > /dev/null 2>&1
if want bench; then
python bench.py > $O/bench_bf16.json 2> $O/bench_bf16.err || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_driver_form.json 2>/dev/null || exit 1
python bench.py --precision f16x2 --no-cpu-baseline > $O/bench_f16x2.json 2> $O/bench_f16x2.err || exit 1
python bench.py --precision fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err || exit 1
python bench.py --precision fp16 --no-cpu-baseline --no-secondary > $O/bench_fp16.json 2>/dev/null || exit 1
for c in vitb8_448 vitl14_518; do python bench.py --precision f16x2 --config $c --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-plain-chain > $O/bench_f16x2_$c.json 2>/dev/null || exit 1; done
python bench.py --precision f16x2 --pairs 8 --steps 60 --no-cpu-baseline --no-secondary --no-plain-chain > $O/bench_f16x2_pairs8.json 2>/dev/null || exit 1
python tools/soak_pipeline.py 20000 > $O/soak_pipeline.txt 2>&1 || exit 1
python tools/soak_pipeline.py 1500 vitb8_448 >> $O/soak_pipeline.txt 2>&1 || exit 1
tail -c 600 $O/bench_bf16.json
fi
if want bench2; then
for b in 2 4 8; do python bench.py --pairs $b --no-cpu-baseline --no-secondary --steps 100 > $O/bench_bf16_pairs$b.json 2>/dev/null || exit 1; done
# the other BASELINE.json configurations (parity-test cases; not the headline)
for c in vitb8_448 vitl14_518 vits14_308 vits16_224; do python bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain > $O/bench_bf16_$c.json 2>/dev/null || exit 1; done
# the reference's shipped default: DINOv2 ViT-S/14 308² with 3x3 log-binned descriptors (config.yaml:17)
python bench.py --config vits14_308 --binned --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_bf16_vits14_308_binned.json 2>/dev/null || exit 1
python bench.py --config vitb8_448 --selection dense --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain > $O/bench_bf16_vitb8_448_dense.json 2>/dev/null || exit 1
python bench.py --config vitl14_518 --precision fp16 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain > $O/bench_fp16_vitl14_518.json 2>/dev/null || exit 1
# several updates in flight: depth sweep of the headline, dispatch rate and per-operator overlap across queues
for k in 1 2 3 4 5; do python bench.py --in-flight $k --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-plain-chain > $O/bench_bf16_in_flight$k.json 2>/dev/null || exit 1; done
VITVS_BENCH_SHARE_GPU=1 VITVS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain > $O/bench_gpus2_gloo_rehearsal.json 2>/dev/null || exit 1
VITVS_BENCH_SHARE_GPU=1 VITVS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain --no-gather > $O/bench_gpus2_gloo_rehearsal_no_gather.json 2>/dev/null || exit 1
VITVS_BENCH_FORCE_DIST=1 python bench.py --steps 100 --no-cpu-baseline --no-secondary --no-plain-chain > $O/bench_rccl_world_of_one.json 2>/dev/null || exit 1
tail -c 300 $O/bench_bf16_pairs8.json
fi
if want prof; then
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-plain-chain > $GRAFT_REPO_ROOT/$O/bench_bf16_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof_trace.err ) || exit 1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_$c.err ) || exit 1
done
# the split-f16 parity mode: kernel stats + traffic of the same command in that precision
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_x2 -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --precision f16x2 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/rocprof_trace_x2.err ) || exit 1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmcx2_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --precision f16x2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmcx2_$c.err ) || exit 1
done
# many-row configuration (8 pairs): kernel trace for the 256-row GEMM tiles and the long attention (ViT-B/8 448)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_b8 -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --config vitb8_448 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/rocprof_trace_b8.err ) || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmcb8_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --config vitb8_448 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmcb8_$c.err ) || exit 1
done
# 8 pairs per update: the 256-row GEMM tiles at 3152 rows and the batched short-sequence attention
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_p8 -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --pairs 8 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/rocprof_trace_p8.err ) || exit 1
# ViT-L/14 518² (configs[4]): 2740 rows, 1370-token attention
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_l -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --config vitl14_518 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-plain-chain > /dev/null 2> $GRAFT_REPO_ROOT/$O/rocprof_trace_l.err ) || exit 1
# keep only the small summaries in the merged output
python tools/trace_summary.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/trace_summary_bf16.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16.csv
python tools/trace_summary.py $(find $O/trace_b8 -name "*kernel_trace.csv" | head -1) > $O/trace_summary_bf16_vitb8_448.txt
cp $(find $O/trace_b8 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16_vitb8_448.csv
python tools/trace_summary.py $(find $O/trace_p8 -name "*kernel_trace.csv" | head -1) > $O/trace_summary_bf16_pairs8.txt
python tools/trace_summary.py $(find $O/trace_l -name "*kernel_trace.csv" | head -1) > $O/trace_summary_bf16_vitl14_518.txt
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv" | head -1) --commit "$COMMIT" --command "rocprofv3 --pmc <counter> -- $CMD" > $O/pmc_traffic.json
python tools/trace_summary.py $(find $O/trace_x2 -name "*kernel_trace.csv" | head -1) > $O/trace_summary_f16x2.txt
cp $(find $O/trace_x2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_f16x2.csv
python tools/pmc_summary.py $(find $O/pmcx2_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmcx2_WRITE_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmcx2_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv" | head -1) --commit "$COMMIT" --command "rocprofv3 --pmc <counter> -- python3 bench.py --precision f16x2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-plain-chain" > $O/pmc_traffic_f16x2.json
python tools/pmc_summary.py $(find $O/pmcb8_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmcb8_WRITE_SIZE -name "*counter_collection.csv" | head -1) --commit "$COMMIT" --command "rocprofv3 --pmc <counter> -- python3 bench.py --config vitb8_448 --steps 5 --warmup 2 ..." > $O/pmc_traffic_vitb8_448.json
rm -rf $O/trace_x2 $O/pmcx2_FETCH_SIZE $O/pmcx2_WRITE_SIZE $O/pmcx2_SQ_VALU_MFMA_BUSY_CYCLES
rm -rf $O/trace $O/trace_b8 $O/trace_p8 $O/trace_l $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmcb8_FETCH_SIZE $O/pmcb8_WRITE_SIZE
fi
if want micro; then
# the probe build of the library (in-kernel stamps; vit-vs_amd/variants/ does not travel with gpurun, so it is built on the box)
mkdir -p vit-vs_amd/variants/probe && make -j8 -C vit-vs_amd/csrc OUT=../variants/probe/libvitvs_hip.so BUILD=build_probe EXTRA=-DVITVS_PROBE > /dev/null 2>&1
python tools/l2_warm_probe.py --precision bf16 > $O/l2_warm_probe.txt 2>&1
python tools/denorm_probe.py > $O/denorm_probe.txt 2>&1
VITVS_ATTN_ONES=1 tools/big_ops attn > $O/attention_ones_variant.txt 2>&1
tools/launch_floor queues > $O/launch_floor_queues.txt 2>&1
tools/op_chain queues > $O/op_chain_queues.txt 2>&1
python tools/vendor_compare.py > $O/vendor.txt 2> $O/vendor.err
tools/launch_floor > $O/launch_floor.txt 2>&1
tools/valu_rate > $O/valu_rate.txt 2>&1
tools/op_chain > $O/op_chain_bf16.txt 2>&1
tools/big_ops > $O/big_ops.txt 2>&1
tools/big_ops mid > $O/big_ops_mid.txt 2>&1
tools/big_ops slices > $O/big_ops_slices.txt 2>&1
VITVS_WEIGHT_MB=600 tools/big_ops > $O/big_ops_cold.txt 2>&1
tools/big_ops sweep > $O/big_ops_k_sweep.txt 2>&1
tools/big_ops attnmid > $O/attention_mid.txt 2>&1
LD_LIBRARY_PATH=vit-vs_amd/variants/probe tools/big_ops fixed > $O/big_ops_fixed_cost.txt 2>&1
LD_LIBRARY_PATH=vit-vs_amd/variants/probe tools/big_ops attn > $O/attention_probe.txt 2>&1
LD_LIBRARY_PATH=vit-vs_amd/variants/probe tools/big_ops occ > $O/attention_occupancy.txt 2>&1
fi
