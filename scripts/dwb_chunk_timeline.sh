#!/bin/bash
# Builds an instrumented libmfm_hip.so (dw_bf16.hip compiled with -DMFM_DWB_STAMP): wave 0 of every 29th workgroup of the
# one-pass weight-gradient launch prints the shader-clock time it spent per chunk waiting for its LDS-DMA, at the barrier,
# issuing the next chunk and in the fragment reads + MFMAs.  Run on the GPU box:
#   MFM_LIB_PATH=scripts/tmp/stampdw/libmfm_hip_stampdw.so python bench.py --dtype bf16 --batch 2048 --steps 1 --warmup 0 --no-cpu-baseline --no-graph
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(dirname "$HERE")"
C="$ROOT/factorized_amd/csrc"
OUT="$ROOT/scripts/tmp/stampdw"
mkdir -p "$OUT"
make -C "$C" -j16 >/dev/null
OBJS=$(ls "$C"/build/*.o | grep -v dw_bf16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$C" -DMFM_DWB_STAMP -c "$C/dw_bf16.hip" -o "$OUT/dw_bf16.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS "$OUT/dw_bf16.o" -o "$OUT/libmfm_hip_stampdw.so"
rm -f "$OUT/dw_bf16.o"
echo built
