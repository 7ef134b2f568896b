#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
IGD_HIP_ALLOW_EXP_BUILD=1 IGD_AMD_LIBDIR=$PWD/igd_amd/libv_qbst python bench.py --no-cpu --no-extra --no-cold --slab-of 8 --steps 5 --warmup 2 > /dev/null 2>$O/qb_stamps.err
python tools/qb_stamps.py gpurun_out/qb_stamps.bin > $O/qb_stamps_bonly.txt 2>&1
