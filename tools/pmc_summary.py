#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/gpu_pmc.sh into one text summary (mean per launch, per kernel)."""
import collections, csv, glob, sys
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
names = {'downdate_f32': 'downdate_f32_mfma', 'downdate_valu': 'downdate_valu', 'w1_mfma': 'w1_mfma', 'panel_gemm': 'panel_gemm',
         'factor_kernel': 'factor', 'pf_auto_step': 'pf_auto_step', 'pf_auto_resample': 'pf_auto_resample', 'pf_auto_scan1': 'pf_auto_scan1', 'downdate_f64': 'downdate_f64_mfma', 'pht_kernel': 'pht', 'gate_kernel': 'gate', 'gate_final': 'gate_final', 'compact_kernel': 'compact'}
for name in ['sq1', 'sq2', 'fetch', 'write', 'tcc', 'grbm']:
    fs = glob.glob(f'{root}/{name}/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(name, 'no file')
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        short = next((v for k, v in names.items() if k in r['Kernel_Name']), None)
        if short:
            agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(name, k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, 'launches', len(next(iter(v.values()))))
