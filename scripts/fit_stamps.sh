#!/bin/bash
# Profiling build of the fit kernel with in-kernel cycle stamps (-DIA3_FIT_STAMPS) into libia3_stamps.so, then
# scripts/fit_stamps.py on it.  Run on the GPU box through gpurun; the shipped libia3.so is not touched.
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO/imageanalysis3_amd/csrc"
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -D'IA3_FOLD_DEPTHS(X)=X(50)' -DIA3_FIT_STAMPS=${IA3_FIT_STAMPS:-1} -c fit.hip -o build/fit_stamps.o
# the objects the Makefile links (stale objects of earlier builds in build/ are not picked up), fit.o replaced
OBJS=$(make -s print-OBJS | grep -v "^build/fit.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libia3_stamps.so build/fit_stamps.o $OBJS -L/opt/rocm/lib -lhipfft -lhiprtc -ldl -Wl,-rpath,/opt/rocm/lib
cd "$REPO"
python3 scripts/fit_stamps.py "$@"
rm -f imageanalysis3_amd/libia3_stamps.so
