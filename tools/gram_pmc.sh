#!/bin/bash
# Gram pass alone at d = 8 and d = 20 (n = m = 16384): VALU instructions per entry (PMC pass) and kernel time (stats pass),
# plus the wall-clock figures gram_only.py prints.  Output: gpurun_out/gram_pmc_d8_d20.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gram_pmc_d8_d20.txt
: > $OUT
for d in 8 20; do
  rm -rf /tmp/gp_pmc /tmp/gp_st
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/gp_pmc -- python3 $R/tools/gram_only.py 16384 16384 $d > /dev/null 2>&1
  python3 - $d >> $OUT <<PY
import csv, glob, collections, sys
d = int(sys.argv[1])
tot = collections.defaultdict(float); disp = set()
for r in csv.DictReader(open(glob.glob('/tmp/gp_pmc/*/*counter_collection.csv')[0])):
    if 'gram_kernel_v3' in r['Kernel_Name']:
        tot[r['Counter_Name']] += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
n = 16384
print("d = %d, n = m = %d: %s dispatches %d" % (d, n, {k: '%.4e' % v for k, v in sorted(tot.items())}, len(disp)))
print("  VALU wave-instructions per entry: %.1f" % (tot['SQ_INSTS_VALU'] * 64.0 / (len(disp) * float(n) * n)))
PY
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp_st -- python3 $R/tools/gram_only.py 16384 16384 $d > /tmp/gp_wall.txt 2>&1
  grep "gram_kernel_v3" $(ls /tmp/gp_st/*/*kernel_stats.csv | head -1) | cut -d, -f1-5 >> $OUT
  grep "^gram n=" /tmp/gp_wall.txt | sed 's/^/  wall clock from Python: /' >> $OUT
done
cat $OUT
