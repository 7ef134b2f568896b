#!/bin/bash
# tools/build_variant_wt.sh <tag> [EXTRA flags] -- like build_variant.sh, but from the WORKING TREE (tracked files as they are now)
tag=$1; shift
rm -rf /tmp/v_$tag && mkdir -p /tmp/v_$tag && git ls-files -z | tar --null -T - -cf - | tar -x -C /tmp/v_$tag
make -C /tmp/v_$tag LIB=$PWD/igd_amd/libv_$tag "EXTRA=$*" $PWD/igd_amd/libv_$tag/libigd_hip.so $PWD/igd_amd/libv_$tag/libigd.so $PWD/igd_amd/libv_$tag/libigd_synth.so > /tmp/mk_$tag.log 2>&1
ls $PWD/igd_amd/libv_$tag
