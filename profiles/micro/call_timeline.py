"""Timeline of ONE library call from a rocprofv3 kernel trace: start, gap to the previous kernel, duration, name.
usage: call_timeline.py s_kernel_trace.csv <kernel-name substring that occurs once per call> [which occurrence, default -2]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void dfd::', '').replace('dfd::', '')
    n = re.sub(r'\(.*', '', n)
    return n[:78]
names = [short(r['Kernel_Name']) for r in rows]
st = [int(r['Start_Timestamp']) for r in rows]; en = [int(r['End_Timestamp']) for r in rows]
idx = [i for i, n in enumerate(names) if sys.argv[2] in n]
i1 = idx[int(sys.argv[3]) if len(sys.argv) > 3 else -2]
j = i1
while j > 0 and st[j] - en[j - 1] < 300000: j -= 1
k = i1
while k + 1 < len(rows) and st[k + 1] - en[k] < 300000: k += 1
t0 = st[j]
print("kernels", k - j + 1, "span ms", (en[k] - t0) / 1e6)
busy = 0
for i in range(j, k + 1):
    gap = (st[i] - en[i - 1]) / 1e3 if i > j else 0
    d = (en[i] - st[i]) / 1e3; busy += d
    print(f"{(st[i]-t0)/1e3:9.1f} us  gap {gap:7.1f}  dur {d:8.1f}  {names[i]}")
print("busy ms", busy / 1e3)
