#!/bin/bash
# Builds the instrumented variants of libmfm_hip.so for scripts/seqb_step_timeline.py: lstm_seq_bf16.hip compiled with
# -DMFM_SEQB_STAMP=k (k = 1..7, one stamp point per build), linked against the objects of the normal build.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(dirname "$HERE")"
C="$ROOT/factorized_amd/csrc"
OUT="$ROOT/scripts/tmp/stampb"
mkdir -p "$OUT"
make -C "$C" -j16 >/dev/null
OBJS=$(ls "$C"/build/*.o | grep -v lstm_seq_bf16.o)
for k in 1 2 3 4 5 6 7; do
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$C" -fno-slp-vectorize \
      -DMFM_SEQB_STAMP=$k -c "$C/lstm_seq_bf16.hip" -o "$OUT/lstm_seq_bf16_$k.o"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS "$OUT/lstm_seq_bf16_$k.o" -o "$OUT/libmfm_hip_stampb$k.so"
    rm -f "$OUT/lstm_seq_bf16_$k.o"
  ) &
done
wait
echo built
