#!/bin/bash
# Builds the launch-clock variant of libmfm_hip.so for scripts/launch_timeline.py (csrc/lstamp.h): the four translation units
# that carry stamps compiled with -DMFM_LAUNCH_STAMP=1, linked against the objects of the normal build.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(dirname "$HERE")"
C="$ROOT/factorized_amd/csrc"
OUT="$ROOT/scripts/tmp/lstamp"
mkdir -p "$OUT"
bash "$C/build.sh" >/dev/null
STAMPED="lstm_seq_small dec_fc1 elementwise plan"
OBJS=""
for o in "$C"/build/*.o; do
  b=$(basename "$o" .o)
  case " $STAMPED " in *" $b "*) ;; *) OBJS="$OBJS $o" ;; esac
done
for b in $STAMPED; do
  EXTRA=""
  [ "$b" = lstm_seq_small ] && EXTRA="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$C" $EXTRA \
    -DMFM_LAUNCH_STAMP=1 ${MFM_LSTAMP_FLAGS:-} -c "$C/$b.hip" -o "$OUT/$b.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $(for b in $STAMPED; do echo "$OUT/$b.o"; done) -o "$OUT/libmfm_hip_lstamp.so"
echo "built $OUT/libmfm_hip_lstamp.so"
