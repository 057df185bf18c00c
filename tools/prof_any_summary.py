"""Developer helper: one per-kernel table from the passes of tools/prof_any.sh TAG (gpurun_out/TAG_stats, gpurun_out/TAG_pmc_*):
mean duration, HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes: the gfx950 correction of MI355X_MICROARCH.md), and the
SQ counters per launch with the derived ratios (LDS conflicts / LDS-active, matrix-core busy).

    python tools/prof_any_summary.py TAG [kernel-name filter ...] > profiles/rNN_TAG_summary.txt
"""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1]
filt = sys.argv[2:]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')


def short(k):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    k = k.split('(')[0]
    return k if len(k) < 90 else k[:87] + '...'


def want(k):
    return not filt or any(f in k for f in filt)


dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, tag + '_stats', '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if want(r['Kernel_Name']):
            dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, tag + '_pmc_*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if want(r['Kernel_Name']):
            ctr[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(set(dur) | set(ctr), key=lambda k: -sum(dur.get(k, [0]))):
    d = dur.get(k, [])
    print(k)
    if d:
        d2 = sorted(d)
        print('   launches %d   mean %.1f us   median %.1f   min %.1f   (total %.2f ms)' % (len(d), sum(d) / len(d), d2[len(d) // 2], d2[0], sum(d) * 1e-3))
    c = {n: sum(v) / len(v) for n, v in ctr.get(k, {}).items()}
    for n in sorted(c):
        print('   %-32s %14.5g  per launch (x%d)' % (n, c[n], len(ctr[k][n])))
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        b = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
        line = '   => HBM traffic %.4g GB per launch (2 x FETCH + WRITE)' % (b * 1e-9)
        if d:
            line += ' = %.2f TB/s over the mean duration' % (b / (sum(d) / len(d) * 1e-6) * 1e-12)
        print(line)
    if c.get('SQ_LDS_IDX_ACTIVE'):
        print('   => LDS bank conflicts / LDS-active cycles = %.3f' % (c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']))
    if c.get('SQ_WAVE_CYCLES'):
        w = c['SQ_WAVE_CYCLES']
        print('   => of wave-cycles: waiting (s_waitcnt / barrier) %.3f, issue-stalled %.3f, issuing %.3f' % (
            c.get('SQ_WAIT_ANY', 0) / w, c.get('SQ_WAIT_INST_ANY', 0) / w, c.get('SQ_ACTIVE_INST_ANY', 0) / w))
    if c.get('GRBM_GUI_ACTIVE') and c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        print('   => matrix-core busy = %.3f of 1024 SIMDs x GRBM_GUI_ACTIVE / 8' % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8)))
    if c.get('GRBM_GUI_ACTIVE') and c.get('SQ_BUSY_CYCLES'):
        print('   => clock estimate: GRBM_GUI_ACTIVE / 8 = %.4g cycles per launch' % (c['GRBM_GUI_ACTIVE'] / 8))
