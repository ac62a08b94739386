"""Timeline of the last traced step of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: per stream
(Queue_Id) the kernels in start order with the idle gap in front of each, the union-busy time of the GPU, the time only
one / both streams are busy, and a per-kernel-class table of (time on the critical stream, exposed gaps).
usage: python tools/timeline.py <rocprof dir> [--full]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_nchw_to_nhwc' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]['Start_Timestamp'])
def S(r): return (int(r['Start_Timestamp']) - t0) / 1e3
def E(r): return (int(r['End_Timestamp']) - t0) / 1e3
def short(n):
    n = n.replace('void ', '').replace('fu::', '').replace('(anonymous namespace)::', '')
    return n.split('(')[0][:44]
queues = sorted(set(r['Queue_Id'] for r in step))
span = max(E(r) for r in step)
print(f"step span {span:.1f} us, {len(step)} launches, queues {queues}")
# union busy
ev = sorted([(S(r), 1) for r in step] + [(E(r), -1) for r in step])
busy = both = 0.0; depth = 0; last = 0.0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: both += t - last
    depth += d; last = t
print(f"GPU busy (union) {busy:.1f} us, idle {span - busy:.1f} us, two kernels in flight {both:.1f} us")
main_q = max(queues, key=lambda q: sum(1 for r in step if r['Queue_Id'] == q))
for q in queues:
    rs = [r for r in step if r['Queue_Id'] == q]
    tot = sum(E(r) - S(r) for r in rs)
    gaps = []
    prev = None
    for r in rs:
        if prev is not None: gaps.append(S(r) - prev)
        prev = E(r)
    print(f"queue {q}: {len(rs)} kernels, kernel time {tot:.1f} us, sum of gaps {sum(g for g in gaps if g > 0):.1f} us"
          f" (median gap {sorted(gaps)[len(gaps)//2] if gaps else 0:.2f})")
if '--full' in sys.argv:
    prev = {}
    for r in step:
        q = r['Queue_Id']
        gap = S(r) - prev.get(q, S(r))
        prev[q] = E(r)
        print(f"q{q} {S(r):8.1f} {E(r)-S(r):7.1f} gap {gap:6.2f}  {short(r['Kernel_Name'])}  wgs={int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X']))}")
# per class on the main queue: kernel time and the gap in front
cls = {}
prev = None
for r in [r for r in step if r['Queue_Id'] == main_q]:
    n = short(r['Kernel_Name'])
    n = n.split('<')[0]
    c = cls.setdefault(n, [0, 0.0, 0.0])
    c[0] += 1; c[1] += E(r) - S(r)
    if prev is not None: c[2] += max(0.0, S(r) - prev)
    prev = E(r)
print(f"main queue {main_q}: class, launches, kernel us, gap-in-front us")
for n, c in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:40s} {c[0]:4d} {c[1]:9.1f} {c[2]:8.1f}")
