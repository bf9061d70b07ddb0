#!/usr/bin/env python3
"""Marginal cost of kernel families inside the (overlapped) label pass: replace chosen C-ABI entry points by no-ops and re-time
`bench.py --profile-pass`.  Outputs are garbage; only the timing is of interest.  usage: skip_probe.py SPEC DEPTH
SPEC: comma list of pw_l4, pw_l3, pw_rest, k2_l4, k2_l3, k2_rest, pyr, prep, c3, pool, bil, label, none"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import mspl_amd
from mspl_amd import _native
lib = _native.lib
spec = set(sys.argv[1].split(','))


def patch(name, pred):
    real = getattr(lib, name)
    def stub(*a):
        return 0 if pred(a) else real(*a)
    setattr(lib, name, stub)


def val(v):
    return v.value if hasattr(v, 'value') else v

pw = lambda a: (('pw_l4' in spec and val(a[6]) == 540) or ('pw_l3' in spec and val(a[6]) == 2160) or
                ('pw_rest' in spec and val(a[6]) not in (540, 2160)))
patch('mspl_conv1x1_fwd', pw)
k2 = lambda a: (('k2_l4' in spec and val(a[6]) == 18) or ('k2_l3' in spec and val(a[6]) == 36 and val(a[3]) == 1) or
                ('k2_rest' in spec and not (val(a[6]) == 18 or (val(a[6]) == 36 and val(a[3]) == 1))))
patch('mspl_eesp_dw_hff_fwd', k2)
if 'pyr' in spec:
    patch('mspl_pyrpool_fused_fwd', lambda a: True)
if 'prep' in spec:
    patch('mspl_pyr_down_prep_fwd', lambda a: True)
if 'c3' in spec:
    patch('mspl_conv3x3_fwd', lambda a: True)
if 'pool' in spec:
    patch('mspl_avgpool3x3s2_fwd', lambda a: True)
    patch('mspl_avgpool3x3s2_psum_fwd', lambda a: True)
if 'bil' in spec:
    patch('mspl_bilinear_fwd', lambda a: True)
if 'label' in spec:
    patch('mspl_label_epilogue_fwd', lambda a: True)
    patch('mspl_label_epilogue_hist_fwd', lambda a: True)
grp = sys.argv[3] if len(sys.argv) > 3 else '1'
sys.argv = ['bench.py', '--profile-pass', '--in-flight', sys.argv[2], '--group', grp, '--steps', '90', '--warmup', '18']
import bench
bench.main()
