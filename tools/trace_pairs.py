"""print the kernel sequence of the LAST vfm_match_mutual_pairs call in gpurun_out/prof_pairs (tools/prof_pairs.sh)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_pairs/pairs_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'l2i8_sumsq' in r['Kernel_Name']]
start = idx[-3]
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    nm = r['Kernel_Name'].replace('vfmm::(anonymous namespace)::', '').replace('void ', '')[:50]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d > 12:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  dur {d:8.1f}  {nm}")
print('total', (int(rows[-1]['End_Timestamp']) - t0) / 1e3)
