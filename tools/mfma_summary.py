"""Developer helper: per-kernel means of the counter passes of tools/mfma_pmc.sh -> CSV on stdout.
Derived columns (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs):
  kernel_cycles = GRBM_GUI_ACTIVE / 8;  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel_cycles);
  mfma_flops = 512 x (MOPS_BF16 + MOPS_F32 ...) is not used: the FLOP/s column comes from the useful FLOPs of the call (bench_configs.secondary)."""
import collections, csv, glob, re, sys
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/mfma_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
        k = re.sub(r'\((?!anonymous).*', '', k)[:90]                      # drop the argument list, keep the template arguments
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
cols = ['SQ_INSTS_MFMA', 'SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES',
        'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_INSTS_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE']
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'dispatches_seen'] + cols + ['kernel_cycles', 'mfma_busy_frac', 'lds_conflict_frac'])
for k in sorted(acc):
    if not any(c in acc[k] for c in ('SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES')) or sum(acc[k].get('SQ_INSTS_MFMA', [0])) == 0:
        continue
    mean = {c: (sum(acc[k][c]) / len(acc[k][c]) if acc[k].get(c) else None) for c in cols}
    kc = mean['GRBM_GUI_ACTIVE'] / 8 if mean['GRBM_GUI_ACTIVE'] else None
    busy = mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * kc) if kc and mean['SQ_VALU_MFMA_BUSY_CYCLES'] is not None else None
    conf = mean['SQ_LDS_BANK_CONFLICT'] / mean['SQ_LDS_IDX_ACTIVE'] if mean['SQ_LDS_IDX_ACTIVE'] else None
    w.writerow([k, max(len(v) for v in acc[k].values())] + ['%.4g' % mean[c] if mean[c] is not None else '' for c in cols] +
               ['%.4g' % kc if kc else '', '%.3f' % busy if busy is not None else '', '%.3f' % conf if conf is not None else ''])
