#!/bin/bash
# step timeline of the fused K2+K3 kernel per wave role, with phases switched off (STAMPS=1 build; see tools/xe_dbg.sh for the bits)
make -C mspl_amd/csrc STAMPS=1 -B build/eesp_exp.o > /dev/null 2>&1; make -C mspl_amd/csrc STAMPS=1 > /dev/null 2>&1
for d in 0 2 4 6 1; do echo "dbg=$d"; MSPL_XE_DBG=$d MSPL_XE_STAMP=1 python tools/bench_ops.py expx 2>&1 | grep "xe stamp" | sed -n "13,14p" | sed 's/(step stamps[^:]*)//' | cut -c1-330; done
