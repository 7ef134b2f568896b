#!/bin/bash
# tools/clustered_ab.sh -- GPU box: the clustered stress database (tools/clustered_probe.py) under every library build
for d in igd_amd/lib igd_amd/libv_*; do
  [ -f $d/libigd_hip.so ] || continue
  echo "== $(basename $d)"; IGD_AMD_LIBDIR=$PWD/$d python tools/clustered_probe.py 2>&1 | tail -1
done
