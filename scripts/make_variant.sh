#!/bin/bash
# scripts/make_variant.sh NAME ['extra -D flags']: copies the library sources to scratch/NAME (for an edit or extra
# defines) and builds them there; run the result with TR_LIBRARY=scratch/NAME/tiny_renderer_amd/lib/libtiny_renderer.so
set -e
cd "$(dirname "$0")/.."
d=scratch/$1
rm -rf "$d" && mkdir -p "$d/tiny_renderer_amd"
cp -r include "$d/"
mkdir -p "$d/tiny_renderer_amd/csrc"
for f in tiny_renderer_amd/csrc/*; do [ -f "$f" ] && cp "$f" "$d/tiny_renderer_amd/csrc/"; done
[ -n "$VARIANT_EDIT" ] && (cd "$d/tiny_renderer_amd/csrc" && eval "$VARIANT_EDIT")
(cd "$d/tiny_renderer_amd/csrc" && make -j4 CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -I../../include -I. -Ibuild -I/opt/rocm/include $2" > build.log 2>&1 || (tail -5 build.log; exit 1))
ls -la "$d/tiny_renderer_amd/lib/libtiny_renderer.so"
