#!/usr/bin/env python3
"""Summarise the PMC passes of tools/gpu_pmc.sh for the probe's own dispatches (the last 4
launches of the conv kernel in each pass)."""
import collections
import csv
import glob
import sys

tag = sys.argv[1]
acc = {}
for f in sorted(glob.glob('gpurun_out/pmc/%s_p*/p_counter_collection.csv' % tag)):
    byc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name'] or 'conv_v2' in r['Kernel_Name']:
            byc[r['Counter_Name']].append(r)
    for c, rs in byc.items():
        sel = rs[-3:]
        acc[c] = sum(float(r['Counter_Value']) for r in sel) / len(sel)
        acc['_grid'] = (sel[0]['Grid_Size'], sel[0]['Workgroup_Size'], sel[0].get('VGPR_Count'), sel[0].get('Accum_VGPR_Count'))
        acc['_ns'] = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / len(sel)
for k in sorted(acc):
    print('%-28s %s' % (k, acc[k] if k.startswith('_') else '%.5g' % acc[k]))
if 'SQ_WAVE_CYCLES' in acc:
    w = acc['SQ_WAVE_CYCLES']
    print('wave-cycle split: wait_any %.1f%%  wait_inst %.1f%%  active %.1f%%' % (
        100 * acc['SQ_WAIT_ANY'] / w, 100 * acc['SQ_WAIT_INST_ANY'] / w, 100 * acc['SQ_ACTIVE_INST_ANY'] / w))
    if acc.get('SQ_INSTS_MFMA'):
        print('non-MFMA VALU per MFMA: %.2f ; LDS instr per MFMA %.2f ; VMEM per MFMA %.3f' % (
            (acc['SQ_INSTS_VALU'] - acc['SQ_INSTS_MFMA']) / acc['SQ_INSTS_MFMA'],
            acc['SQ_INSTS_LDS'] / acc['SQ_INSTS_MFMA'], acc['SQ_INSTS_VMEM'] / acc['SQ_INSTS_MFMA']))
if 'FETCH_SIZE' in acc:
    print('FETCH_SIZE KB %.0f (x2 for wide streams per guide) WRITE_SIZE KB %.0f' % (acc['FETCH_SIZE'], acc['WRITE_SIZE']))
