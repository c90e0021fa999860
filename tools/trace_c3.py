"""kernel sequence of ONE C3 registration in the mode given (default int8-top2): python tools/trace_c3.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'prep_chunk_kernel' in r['Kernel_Name']]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    nm = r['Kernel_Name'].replace('vfmm::(anonymous namespace)::', '').replace('(anonymous namespace)::', '').replace('void ', '')[:56]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  dur {d:8.1f}  {nm}")
print('total', (int(rows[-1]['End_Timestamp']) - t0) / 1e3)
