"""The dominant kernel's launches of bench.py's roofline leg in a rocprofv3 kernel trace (CSV):
python tools/dominant_from_trace.py run_kernel_trace.csv [n_timed=50]
The leg is the longest run of consecutive dispatches of the bf16 3x3 kernel with the producer's affine (conv3x3_kernel<bf16, true>,
one persistent 512-thread block per CU); its last n_timed launches are the ones bench.py brackets with HIP events."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_timed = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def is_dom(r):
    # (rocprofv3 demangles the affine instance badly: "conv3x3_kernel<bool _Accum, bool, E>"; the other one stays mangled)
    # (round 5: conv3x3_cs_kernel<bf16, ACT = 1 affine + ReLU, MODE = 1 statistics> - mangled 'conv3x3_cs_kernelIDF16bLi1ELi1ELb0E')
    nm = r['Kernel_Name']
    # (rocprofv3 demangles "conv3x3_cs_kernel<__bf16, 1, 1, false>" as "conv3x3_cs_kernel<bool _Accum, int, E, 1, false>": the ACT
    #  argument is lost, the MODE = 1 (statistics) and GA = false arguments survive)
    # (since the 64-channel-tile flag: "<bool _Accum, int, E, 1, false, false>" / 'Li1ELi1ELb0ELb0E')
    cs = 'conv3x3_cs_kernel' in nm and ('E, 1, false, false>' in nm or 'Li1ELi1ELb0ELb0E' in nm or 'E, 1, false>' in nm or nm.endswith('Li1ELi1ELb0EEEvN6mpn_c35GroupE'))
    return (cs or 'conv3x3_kernel<' in nm) and int(r['Grid_Size_X']) == 256 * 512


best, cur = [], []
for r in rows:
    if is_dom(r):
        cur.append(r)
    else:
        if len(cur) > len(best):
            best = cur
        cur = []
if len(cur) > len(best):
    best = cur
timed = best[-n_timed:]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in timed]
span = (int(timed[-1]['End_Timestamp']) - int(timed[0]['Start_Timestamp'])) / 1e3 / len(timed)
print(f"dominant kernel (3x3 128 -> 128 forward with affine + statistics), the {len(timed)} timed launches of bench.py's roofline leg "
      f"(run of {len(best)} consecutive dispatches on [32,128,128,128]): mean {sum(d) / len(d):.1f} us (min {min(d):.1f}, "
      f"max {max(d):.1f}); start-to-end span per launch {span:.1f} us")
