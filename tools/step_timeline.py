"""Timeline of the LAST benchmark step in a rocprofv3 kernel-trace CSV of bench.py: kernels per hardware queue, time-sorted,
from the third-last Gram launch (K(xi,xi) of the prediction) to the end.  Usage: step_timeline.py trace.csv [max_rows]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
gi = [i for i, r in enumerate(rows) if 'gram_kernel' in r['Kernel_Name']]
seg = rows[gi[-2]:] if '--predict' in sys.argv else rows[gi[-3]:]
t0 = int(seg[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in seg)
def short(n):
    m = re.search(r'(gemm_f64_kernel(?:_v2)?<[^>]*>|\w+_kernel\w*(?:<[^>]*>)?)', n); return m.group(1) if m else n[:40]
print("step span %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
busy = {}
for r in seg:
    q = r['Queue_Id']; busy[q] = busy.get(q, 0) + int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("busy per queue (ms):", {q: round(v / 1e6, 3) for q, v in busy.items()})
for r in seg[: int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 400]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print("q%s %-50s start %8.3f dur %7.3f wg %d" % (r['Queue_Id'], short(r['Kernel_Name']), (s - t0) / 1e6, (e - s) / 1e6,
                                                     int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // (int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))))
