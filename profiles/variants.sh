#!/bin/bash
# Builds A/B variants of libdbde_hip.so under profiles/variants/<name>/ (git-ignored, shipped to the GPU box)
# and the Python-free timing driver profiles/abbench.  Usage: profiles/variants.sh name="-DFLAG ..." ...
# `r01` builds the kernels of the round-1 head (git da16388) as the reference point.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 profiles/abbench.cpp -o profiles/abbench -ldl
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}; [ "$flags" == "$spec" ] && flags=""
    out=$ROOT/profiles/variants/$name
    mkdir -p $out
    if [ "${name#rev_}" != "$name" ]; then   # rev_<git revision>: the library as it was at that revision (same-box A/B against history)
        tmp=$(mktemp -d)
        git archive ${name#rev_} dbde-video-cpp_amd/csrc include | tar -x -C $tmp
        make -s -C $tmp/dbde-video-cpp_amd/csrc
        cp $tmp/dbde-video-cpp_amd/libdbde_hip.so $out/
        rm -rf $tmp
    elif [ "$name" == "r01" ]; then
        tmp=$(mktemp -d)
        git archive da16388 dbde-video-cpp_amd/csrc include | tar -x -C $tmp
        make -s -C $tmp/dbde-video-cpp_amd/csrc
        cp $tmp/dbde-video-cpp_amd/libdbde_hip.so $out/
        rm -rf $tmp
    else
        make -s -B -C dbde-video-cpp_amd/csrc OUT=$out EXTRA="$flags" $out/libdbde_hip.so
    fi
    echo "built $name ($flags)"
done
