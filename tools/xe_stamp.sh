#!/bin/bash
# step timeline of the fused K2+K3 kernel (s_memrealtime stamps per wave; STAMPS=1 build).  GPU box: bash tools/xe_stamp.sh
# prints, per launch of tools/bench_ops.py expx (level-3 and level-4 shape, batch K2_N or 16): averages and maxima over the waves of
# start, the two prologue barriers, every step's "work done" time and the end.
# (a STAMPS=1 build goes to mspl_amd/lib/libmspl_hip_stamps.so: the product library is not touched)
make -C mspl_amd/csrc STAMPS=1 -j8 > /dev/null 2>&1
MSPL_HIP_LIB=$PWD/mspl_amd/lib/libmspl_hip_stamps.so MSPL_XE_STAMP=1 python tools/bench_ops.py expx 2>&1 | grep "xe stamp\|^---" | cut -c1-420
