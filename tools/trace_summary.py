"""Summarise the last training step in a rocprofv3 kernel trace CSV: python tools/trace_summary.py trace.csv [n_last]"""
import collections
import csv
import sys

KEYS = ['slab_compact', 'l2_loss', 'l2_partial', 'l2_final', 'conv3x3', 'pw_gemm', 'bn_bwd_reduce', 'bn_bwd_apply', 'bn_finalize', 'bn_bwd_finalize', 'bn_act', 'dwconv_fwd', 'dwconv_wgrad',
        'dwconv_dgrad', 'dwconv_bwd', 'conv_wgrad', 'conv_mfma', 'reduce_partials', 'stem_wgrad', 'stem_fwd', 'pack_weights',
        'bilinear_up_fwd', 'bilinear_up_bwd', 'slice_copy', 'slice_affine_store', 'head_bwd', 'head_fwd', 'loss_kernel', 'loss_finalize', 'adam', 'add_inplace',
        'sumpool', 'bn_stats', 'copyBuffer', 'decode',
        'retina_match', 'retina_loss', 'retina_nms', 'retina_', 'patchify', 'prn_loss', 'prn_', 'transpose_cast', 'cast_kernel', 'bias_relu', 'l2_loss', 'axpy']


def short(nm):
    for k in KEYS:
        if k in nm:
            return k
    return nm[:30]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with adam + pack_weights: cut at the last two adam dispatches
adam = [i for i, r in enumerate(rows) if 'adam_apply' in r['Kernel_Name']]
if len(adam) >= 2:
    last = rows[adam[-2] + 1:adam[-1] + 1]
    # the packs of the previous step trail its adam; drop them from the head
    while last and 'pack_weights' in last[0]['Kernel_Name']:
        last.pop(0)
else:
    last = rows
t0 = int(last[0]['Start_Timestamp'])
agg = collections.OrderedDict()
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(short(r['Kernel_Name']), []).append(d)
span = (int(last[-1]['End_Timestamp']) - t0) / 1e3
tot = sum(sum(v) for v in agg.values())
print(f"dispatches {len(last)}  span {span:.0f} us  busy {tot:.0f} us  gaps {span - tot:.0f} us")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:18s} n={len(v):3d} sum={sum(v):8.1f} ({100 * sum(v) / tot:4.1f}%) max={max(v):7.1f}  first: {[round(x) for x in v[:16]]}")
