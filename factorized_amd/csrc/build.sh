#!/bin/bash
# Build libmfm_hip.so for gfx950 in-tree (cross-compiles without a GPU); incremental through the Makefile.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
make -C "$HERE" -f "$HERE/Makefile" -j"$(nproc)" "$@"
