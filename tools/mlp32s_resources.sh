#!/bin/bash
# Register / scratch figures of the split-fp16 MLP kernel alone (1.5 s instead of the whole library's 35 s).  Usage: tools/mlp32s_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/../sigmarl_amd/csrc" || exit 1
cat > _mlp_only.hip <<'EOS'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#define MLP32_ROWS 64
#define MLP32_MAX_LAYERS 4
#define MLP32_H 256
#define MLP32_TS_ARG
#define MLP32_TS(k) do { } while (0)
#include "sigmaenv_mlp32s.inc"
EOS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -disable-machine-licm --cuda-device-only "$@" -S -Rpass-analysis=kernel-resource-usage -o /tmp/_mlp32s.s _mlp_only.hip 2>&1 \
  | grep -E "error|Function Name|VGPRs|Scratch|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//'
rm -f _mlp_only.hip
