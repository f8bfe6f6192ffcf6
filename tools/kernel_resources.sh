#!/bin/bash
# Register / scratch / spill figures of the HIP kernels (compiler remarks of the product build flags).  Usage: tools/kernel_resources.sh [name-regex]
cd "$(dirname "$0")/../sigmarl_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -disable-machine-licm ${EXTRA_FLAGS} -Rpass-analysis=kernel-resource-usage -o /tmp/_kr.so sigmaenv.hip 2>&1 \
  | grep -E "error|Function Name|SGPRs|VGPRs|ScratchSize|Spill|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' \
  | awk -v pat="${1:-.}" '/Function Name/ {show = ($0 ~ pat)} show || /error/ {print}'
