#!/bin/bash
# An experiment build of the library with extra compile-time definitions, next to the product build:
#   bash tools/build_variant.sh sb256 -DMG_SHADE_BLOCK=256     ->  mitsuba-renderer_amd/libmtsgpu_sb256.so
# Select it at run time with MTSGPU_LIB=<path> (mitsuba-renderer_amd/__init__.py).
set -e
name=$1; shift
cd "$(dirname "$0")/../mitsuba-renderer_amd/csrc"
# the same stamp as csrc/Makefile: a variant answers mtsgpu_source_hash() with the hash of the sources it was built from
K="sampler.hip film.hip trace.hip shade.hip measure.hip"
hash=$(cat api.cpp group.cpp $K kdbuild.cpp flatten.cpp serialized.cpp host.h ctx.h kernels.h kdevice.h sampler.h devmath.h ../../include/mtsgpu.h | sha256sum | cut -c1-16)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -pthread -DMTSGPU_SOURCE_HASH="\"$hash\"" "$@" \
    -x hip -shared api.cpp group.cpp $K kdbuild.cpp flatten.cpp serialized.cpp stamp.cpp -o ../libmtsgpu_$name.so -lz -ldl
ls -la ../libmtsgpu_$name.so
