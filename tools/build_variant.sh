#!/bin/bash
# An A/B build of ONE translation unit with extra -D flags, linked with the current objects of the others:
#   tools/build_variant.sh <tag> <unit> "<flags>"   ->  rails_amd/_ab/librails_amd_<tag>.so   (select it with RAILS_AMD_LIBRARY)
set -e
cd "$(dirname "$0")/../rails_amd/csrc"
tag=$1; unit=$2; flags=$3
mkdir -p ../_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -Wall -Wno-unused-function $flags -c $unit.hip -o /tmp/ab_${tag}_$unit.o
objs=""; for o in *.o; do [ "$o" = "$unit.o" ] && objs="$objs /tmp/ab_${tag}_$unit.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../_ab/librails_amd_$tag.so
echo built ../_ab/librails_amd_$tag.so
