#!/bin/bash
# A/B of fit-kernel builds on the GPU box: libia3.so as shipped against a build of fit.hip with other flags
# (e.g. AB_FLAGS="-DIA3_FIT_LB=3"), same seeds, tables compared.  usage: scripts/ab_fit2.sh <tag> "<flags>" [ab_fit2.py args]
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; FLAGS=$2; shift 2
cd "$REPO/imageanalysis3_amd/csrc"
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -D'IA3_FOLD_DEPTHS(X)=X(50)' $FLAGS -c fit.hip -o build/fit_$TAG.o
OBJS=$(make -s print-OBJS | grep -v "^build/fit.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libia3_$TAG.so build/fit_$TAG.o $OBJS -L/opt/rocm/lib -lhipfft -lhiprtc -ldl -Wl,-rpath,/opt/rocm/lib
cd "$REPO"
python3 scripts/ab_fit2.py "$@"
IA3_LIB_PATH=$REPO/imageanalysis3_amd/libia3_$TAG.so python3 scripts/ab_fit2.py "$@"
rm -f imageanalysis3_amd/libia3_$TAG.so
