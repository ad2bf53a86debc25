#!/bin/bash
# development only: recompile csrc/awq_midm_cdna4.hip alone with extra flags (e.g. -DAWQ_ENABLE_PROBES -DAWQ_MIDM_DEV_ONE) and relink the library.
# `python -m llm_awq_amd.build` afterwards restores the product build (the source is touched so that it is recompiled).
set -e
cd "$(dirname "$0")/.."
OBJ=llm_awq_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=16 "$@" \
  -c llm_awq_amd/csrc/awq_midm_cdna4.hip -o $OBJ/awq_midm_cdna4.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.hip.o -o llm_awq_amd/lib/libawq_cdna4.so
touch llm_awq_amd/csrc/awq_midm_cdna4.hip
echo "dev library linked with: $*"
