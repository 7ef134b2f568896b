#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
python tools/prep.py > /dev/null 2>&1
IGD_HIP_SPLIT_ONE=1 IGD_HIP_ALLOW_EXP_BUILD=1 IGD_AMD_LIBDIR=$PWD/igd_amd/libv_spst python bench.py --no-cpu --no-extra --no-cold --shuffled --steps 5 --warmup 2 > /dev/null 2>$O/sp_stamps.err
python tools/sp_stamps.py gpurun_out/qb_stamps.bin fine > $O/fine_stamps.txt 2>&1
