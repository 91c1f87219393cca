#!/usr/bin/env bash
# Builds libenlsip_gn.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU present.
# ENLSIP_GN_OUT names another output file (variants: __graft_entry__.build()); extra arguments go to hipcc.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$here/lib"
out="${ENLSIP_GN_OUT:-$here/lib/libenlsip_gn.so}"
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC \
    -I"$here/../include" -o "$out" "$here/csrc/enlsip_gn.hip" -ldl "$@"
