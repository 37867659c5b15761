#!/bin/bash
# development: build_variants/<name>.so = the library with extra -D flags (A/B on one box: WCSPH_LIB=... python tools/tile_bench.py)
# usage: tools/build_variant.sh <name> [flags...]
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
mkdir -p $R/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -Wno-unused-value "$@" -I $R/include -o $R/build_variants/$N.so $R/wcsph_amd/csrc/wcsph_hip.hip
