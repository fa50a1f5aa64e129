"""Summarise a rocprofv3 kernel-trace CSV of tools/potrf_only.py: per-kernel totals per queue and the
main-stream trailing updates of the second factorisation (diagnostic tool)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
gi = [i for i, r in enumerate(rows) if 'gram_kernel' in r['Kernel_Name']]
seg = sorted(rows[gi[-1] + 1:], key=lambda r: int(r['Start_Timestamp']))
t0 = int(seg[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in seg)
print("potrf span ms %.2f  kernels %d" % ((t1 - t0) / 1e6, len(seg)))
def short(n):
    m = re.search(r'(gemm_f64_kernel(?:_v2)?<[^>]*>|\w+_kernel\w*)', n); return m.group(1) if m else n[:30]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    k = (short(r['Kernel_Name']), r['Queue_Id']); d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    agg[k][0] += 1; agg[k][1] += d
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("%-50s q%s n=%4d total %.2f ms avg %.1f us" % (k[0], k[1], a[0], a[1], 1e3 * a[1] / a[0]))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for r in seg[:nshow]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print("q%s %-45s start %8.3f dur %7.3f grid %d" % (r['Queue_Id'], short(r['Kernel_Name']), (s - t0) / 1e6, (e - s) / 1e6, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])))

# chain time per 128 columns: start-to-start distance of consecutive diagonal-block kernels
ps = [int(r['Start_Timestamp']) for r in seg if 'potf2_inv_kernel' in r['Kernel_Name']]
if len(ps) > 2:
    import statistics
    dd = [(b - a) / 1e3 for a, b in zip(ps[:-1], ps[1:])]
    print("chain: %d diagonal blocks, start-to-start us: median %.1f mean %.1f min %.1f max %.1f" % (len(ps), statistics.median(dd), sum(dd) / len(dd), min(dd), max(dd)))
