#!/usr/bin/env bash
# Builds libenlsip_gn.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU present.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$here/lib"
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC \
    -I"$here/../include" -o "$here/lib/libenlsip_gn.so" "$here/csrc/enlsip_gn.hip" -ldl "$@"
