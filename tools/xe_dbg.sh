#!/bin/bash
# phase-skipping timings of the fused K2+K3 kernel (MSPL_XE_DBG bits: 1 no depthwise stage, 2 no MFMAs, 4 no staging, 8 no residual)
for d in 0 1 2 3 4 7 8 15; do echo "dbg=$d"; MSPL_XE_DBG=$d python tools/bench_ops.py exp 2>&1 | grep -o "^L. s1 n=[0-9]* \|fused *[0-9.]* us" | paste - - ; done
