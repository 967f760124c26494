"""Summarise the last training step of a rocprofv3 rocpd database (default output of rocprofv3 7.x):
python tools/rocpd_summary.py run_results.db"""
import collections
import sqlite3
import sys

sys.path.insert(0, 'tools')
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y from {kd} d join {sym} s on d.kernel_id = s.id order by d.start"))
KEYS = ['bn_bwd_reduce', 'bn_bwd_apply', 'bn_finalize', 'bn_bwd_finalize', 'bn_act', 'dwconv_fwd', 'dwconv_wgrad',
        'dwconv_dgrad', 'conv_wgrad', 'conv_mfma', 'reduce_partials', 'stem_wgrad', 'stem_fwd', 'pack_weights',
        'bilinear_up_fwd', 'bilinear_up_bwd', 'head_bwd', 'head_fwd', 'loss_kernel', 'loss_finalize', 'adam', 'add_inplace',
        'sumpool', 'bn_stats', 'copyBuffer', 'decode', 'render']


def short(nm):
    for k in KEYS:
        if k in nm:
            return k
    return nm[:30]


adam = [i for i, r in enumerate(rows) if 'adam_apply' in r[0]]
last = rows[adam[-2] + 1:adam[-1] + 1] if len(adam) >= 2 else rows
while last and 'pack_weights' in last[0][0]:
    last.pop(0)
agg = collections.OrderedDict()
for r in last:
    agg.setdefault(short(r[0]), []).append((r[2] - r[1]) / 1e3)
span = (last[-1][2] - last[0][1]) / 1e3
tot = sum(sum(v) for v in agg.values())
print(f"dispatches {len(last)}  span {span:.0f} us  busy {tot:.0f} us  gaps {span - tot:.0f} us")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:18s} n={len(v):3d} sum={sum(v):8.1f} ({100 * sum(v) / tot:4.1f}%) max={max(v):7.1f}  first: {[round(x) for x in v[:16]]}")
