"""Summarise a rocprofv3 --pmc pass of tools/conv_modes.py: per kernel name, median of each counter and ratios."""
import csv, glob, sys, statistics as st
acc = {}
for fn in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'k_conv3x3' not in r['Kernel_Name']:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void fu::', '')[:40] + f" grid={r.get('Grid_Size','')}"
        acc.setdefault(k, {}).setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
        acc[k][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for k, cs in sorted(acc.items()):
    med = {c: st.median(v.values()) for c, v in cs.items()}
    line = f"{k:60s}"
    wc = med.get('SQ_WAVE_CYCLES')
    for c, v in sorted(med.items()):
        line += f" {c.replace('SQ_','')}={v:.3g}"
        if wc and c != 'SQ_WAVE_CYCLES' and c.startswith('SQ_') and 'BUSY' not in c:
            line += f"({v / wc:.2f})"
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in med and 'SQ_BUSY_CU_CYCLES' in med:
        line += f" | mfma_busy/4cu={med['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * med['SQ_BUSY_CU_CYCLES']):.3f}"
    if 'SQ_LDS_IDX_ACTIVE' in med and 'SQ_BUSY_CU_CYCLES' in med:
        line += f" lds_active/cu={med['SQ_LDS_IDX_ACTIVE'] / med['SQ_BUSY_CU_CYCLES']:.3f}"
    print(line)
