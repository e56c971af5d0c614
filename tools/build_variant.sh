#!/bin/bash
# usage (build container, repo root): tools/build_variant.sh <git revision | WORK> <name> [extra compiler flags]
# Builds the library of a revision (or of the working tree) as build_ab/<name>.so for same-box A/B runs (tools/ab.sh loads
# the variants through GZ_LIBRARY; build_ab/ travels to the GPU box but is not tracked).
set -e
rev=$1; name=$2; shift 2
R=$PWD
mkdir -p build_ab
if [ "$rev" = WORK ]; then src=$R/genz-tokenize_amd/csrc; inc=$R/include
else
  tmp=$(mktemp -d); git archive $rev genz-tokenize_amd/csrc include | tar -x -C $tmp
  src=$tmp/genz-tokenize_amd/csrc; inc=$tmp/include
fi
cd $src
id=$( (cat gz_kernels.hip gz_api.cpp gz_tables.cpp gz_host_api.cpp *.h *.inc $inc/genz_tokenize.h; echo "gfx950 -O3 $*") | sha256sum | cut -c1-16)   # sources AND flags: a variant never shares cache files with another build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-align-mismatch \
  -DGZ_BUILD_ID="\"$id\"" "$@" -x hip gz_kernels.hip -x hip gz_api.cpp -x hip gz_tables.cpp -x hip gz_host_api.cpp -shared -o $R/build_ab/$name.so -ldl
echo "built build_ab/$name.so ($rev, id $id)"
