#!/bin/bash
# Experiment build of the HIP library into xbuild/ (git-ignored, travels to the GPU box):
#   tools/build_exp.sh NAME [extra hipcc flags, e.g. -DWRENC_POOL_MIN_TLG=5]
# then  WRENC_GPU_LIB=xbuild/NAME.so python tools/fill_probe.py ...
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1
shift
mkdir -p xbuild
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -sink-insts-to-avoid-spills=1 -mllvm -disable-machine-licm \
    "$@" -o "xbuild/$name.so" wrenc_amd/csrc/wrenc_gpu.hip 2>&1 | grep -E " error|error:" || true
echo "$(git rev-parse --short HEAD)$(git diff --quiet || echo +dirty) $*" > "xbuild/$name.flags"
ls -la "xbuild/$name.so"
