#!/bin/bash
# A/B builds of the library with extra compile flags:  tools/ab_build.sh <name> <flags...>  -> admp_amd/lib/ab_<name>.so
# (run a tool against it with ADMP_HIP_LIB=$PWD/admp_amd/lib/ab_<name>.so; the default library is left alone)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
objs=""
for f in $(python3 -c "import sys; sys.path.insert(0, '$root'); from admp_amd import build; print(' '.join(build.SOURCES))"); do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -fno-slp-vectorize "$@" -c $root/admp_amd/csrc/$f -o $tmp/${f%.hip}.o &
  objs="$objs $tmp/${f%.hip}.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/admp_amd/lib/ab_$name.so $objs -L/opt/rocm/lib -lrocfft -lhiprtc -Wl,-rpath,/opt/rocm/lib
rm -rf $tmp
echo built $root/admp_amd/lib/ab_$name.so
