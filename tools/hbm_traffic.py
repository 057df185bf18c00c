"""Developer helper: HBM bytes per launch of the headline kernels from the two PMC passes of tools/profile_round.sh
(gpurun_out/pmc_FETCH_SIZE_TAG, pmc_WRITE_SIZE_TAG) -> profiles/hbm_traffic.json (read by bench.py for roofline.traffic).
Counters are in KiB; on gfx950 FETCH_SIZE counts half of the fetched bytes (x2, /opt/skills/guides/MI355X_MICROARCH.md)."""
import csv, glob, collections, json, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
val = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, 'gpurun_out', 'pmc_%s_%s' % (c, tag), '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name']
            s = next((n for n in ('fd_residual', 'spec_xpass', 'spec_ypass', 'spec_rowmarch') if n in kn), None)          # spec_xpass also matches spec_xpass_split_kernel
            if s == 'spec_ypass' and re.search(r'spec_ypass_kernel<\d+, \w+, true', kn):        # <N, TF, FUSE_FD = true, SEGP>
                s = 'both_rowpass_f64_forward'                  # spec_ypass_kernel<N, TF, FUSE_FD = true>: the fused row pass of the float64-forward mode
            if s == 'spec_rowmarch':
                s = 'both_rowpass'                              # the marching form of the fused row pass (the headline's)
            if s == 'spec_xpass' and 'double' in kn:
                s = 'spec_xpass_f64_forward' 
            if s and r['Counter_Name'] == c:
                acc[s].append(float(r['Counter_Value']))
    val[c] = {k: sum(v) / len(v) for k, v in acc.items()}
import subprocess
out = {k: (2 * val['FETCH_SIZE'][k] + val['WRITE_SIZE'][k]) * 1024 for k in val['FETCH_SIZE']}
try:
    commit = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = ''
out['_source'] = dict(tag=tag, commit_at_summary=commit, passes='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one counter per pass) over bench.py --steps 3; '
                      'bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, averaged over launches')
json.dump(out, open(os.path.join(root, 'profiles', 'hbm_traffic.json'), 'w'), indent=1)
print(out)
