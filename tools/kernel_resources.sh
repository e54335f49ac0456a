#!/bin/bash
# Per-kernel register / scratch / occupancy table for the HIP sources (cross-compiles for gfx950, no GPU needed).
cd "$(dirname "$0")/../vit-vs_amd/csrc" || exit 1
for f in "${@:-gemm attention correspond servo elementwise}"; do for g in $f; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form -c $g.hip -o /tmp/_kr.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' \
 | awk '/Function Name/{n=$3} /VGPRs:/{v=$2} /AGPRs:/{a=$2} /ScratchSize/{s=$4} /Occupancy/{o=$4} /LDS Size/{print n, "vgpr="v, "agpr="a, "scratch="s, "occ="o, "lds="$5}' \
 | c++filt | sed -e 's/vitvs:://g' | cut -c1-150
done; done
