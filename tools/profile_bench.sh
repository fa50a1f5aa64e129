#!/bin/bash
# Profile passes of the headline benchmark (run on the GPU box through gpurun): `tools/profile_bench.sh v24 r5` writes
# gpurun_out/prof_r5/bench_*_v24_*; copy what is to be judged into profiles/r5/.
#   1. rocprofv3 --kernel-trace --stats        -> bench_n32768_m50000_<V>_kernel_stats.csv (+ the bench line printed under the profiler)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE -> per-kernel HBM-side traffic (separate passes, as the guide prescribes)
#   3. rocprofv3 --pmc MFMA-busy counters      -> per-kernel MFMA utilisation
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RND=${2:-r5}
OUT=$R/gpurun_out/prof_$RND
mkdir -p $OUT
V=${1:-v24}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-live-pmc > $OUT/bench_${V}_under_rocprofv3.log 2>&1
cp $(ls /tmp/p_stats/*/*kernel_stats.csv | head -1) $OUT/bench_n32768_m50000_${V}_kernel_stats.csv
echo "stats done"; touch /tmp/gpmp_stats_pass_done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-kernel-events --no-live-pmc > /dev/null 2>&1
  lc=$(echo $c | tr 'A-Z' 'a-z')
  python3 $R/tools/pmc_by_kernel.py $(ls /tmp/p_$c/*/*counter_collection.csv | head -1) $c > $OUT/bench_${V}_pmc_${lc}_by_kernel.csv
  echo "$c done"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfma -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-kernel-events --no-live-pmc > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
path = glob.glob('/tmp/p_mfma/*/*counter_collection.csv')[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
    tot[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value']); disp[r['Kernel_Name']].add(r['Dispatch_Id'])
cols = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F64', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE']
w = csv.writer(open('$OUT/bench_${V}_pmc_mfma_by_kernel.csv', 'w'))
w.writerow(['kernel', 'dispatches'] + cols + ['mfma_busy_frac = BUSY / (GRBM_GUI_ACTIVE / 8 * 1024)'])
for k in sorted(tot, key=lambda k: -tot[k]['GRBM_GUI_ACTIVE']):
    t = tot[k]; act = t['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0
    w.writerow([k, len(disp[k])] + ['%.6e' % t[c] for c in cols] + ['%.4f' % (t['SQ_VALU_MFMA_BUSY_CYCLES'] / act if act else 0.0)])
PY
echo "mfma done"
