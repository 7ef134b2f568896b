#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
IGD_HIP_ALLOW_EXP_BUILD=1 IGD_AMD_LIBDIR=$PWD/igd_amd/libv_tlst python3 tools/length_one.py 100000 200000 100000 5 > $O/tail_stamps.txt 2>&1
python tools/tail_stamps.py gpurun_out/tail_stamps.bin >> $O/tail_stamps.txt 2>&1
