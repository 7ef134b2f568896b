#!/bin/bash
# tools/build_variant.sh <tag> <commit> [EXTRA flags] -- build the libraries of a commit into igd_amd/libv_<tag> (for tools/variants.sh A/B runs)
tag=$1; commit=$2; shift 2
rm -rf /tmp/v_$tag && mkdir -p /tmp/v_$tag && git archive $commit | tar -x -C /tmp/v_$tag
make -C /tmp/v_$tag LIB=$PWD/igd_amd/libv_$tag "EXTRA=$*" $PWD/igd_amd/libv_$tag/libigd_hip.so $PWD/igd_amd/libv_$tag/libigd.so $PWD/igd_amd/libv_$tag/libigd_synth.so > /tmp/mk_$tag.log 2>&1
ls $PWD/igd_amd/libv_$tag
