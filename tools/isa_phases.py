"""Instruction mix of a kernel's ISA between consecutive s_barrier instructions: python tools/isa_phases.py file.s <kernel-substring>"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(key.split()[-1]) or (l.startswith("_Z") and key in l.split(":")[0]))
seg, cur = [], {"from": start}
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and "lds" in op): return "lds_dma"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "vmem_load"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "vmem_store"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"
i = start + 1
out = []
cnt = {}
while i < len(lines):
    l = lines[i].strip()
    if l.startswith(".Lfunc_end"):
        out.append((i, dict(cnt))); break
    if l and not l.startswith(";") and not l.startswith(".") and not l.endswith(":"):
        op = l.split()[0]
        c = cls(op)
        cnt[c] = cnt.get(c, 0) + 1
        if c == "barrier":
            out.append((i, dict(cnt))); cnt = {}
    if l.endswith(":") and l.startswith(".LBB"):
        cnt["label"] = cnt.get("label", 0) + 1
    i += 1
keys = ["mfma", "valu", "ds_read", "ds_write", "lds_dma", "vmem_load", "vmem_store", "salu", "waitcnt", "branch", "label"]
print("line    " + " ".join(f"{k:>10s}" for k in keys))
for ln, c in out:
    print(f"{ln:6d}  " + " ".join(f"{c.get(k, 0):10d}" for k in keys))
