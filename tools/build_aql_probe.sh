#!/bin/bash
# tools/build_aql_probe.sh -> tools/aql_probe + tools/step_kernels.hsaco (the step kernels as a bare gfx950 code object, same
# flags as the product build) for tools/aql_probe.cpp.  Cross-compiles without a GPU.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -DENV2048_BUILDING -I$R/include -I$R/simulate_2048_amd/csrc -Wno-unused-function \
    -mllvm -amdgpu-kernarg-preload-count=16 --offload-device-only --no-gpu-bundle-output "$@" \
    $R/simulate_2048_amd/csrc/step_kernels.hip -o $R/tools/${HSACO:-step_kernels}.hsaco
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only --no-gpu-bundle-output $R/tools/aql_aux.hip -o $R/tools/aql_aux.hsaco
$HIPCC -O2 -std=c++17 -I$R/include -I$R/simulate_2048_amd/csrc $R/tools/aql_probe.cpp -o $R/tools/aql_probe \
    -L$R/simulate_2048_amd/lib -lenv2048 -Wl,-rpath,'$ORIGIN/../simulate_2048_amd/lib' -L/opt/rocm/lib -lhsa-runtime64
echo built tools/aql_probe tools/step_kernels.hsaco tools/aql_aux.hsaco
