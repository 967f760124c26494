"""profiles/<tag>_dominant_kernel_traffic.json from the counter passes of tools/pmc_passes.sh <tag> '...' tools/one_conv.py fwd 128 128 128 3:
python tools/make_traffic_json.py <tag> <pmc dir tag>   (HBM bytes per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE doubled on gfx950)"""
import collections, csv, glob, json, sys

tag, pm = sys.argv[1], sys.argv[2]
vals = {}
dur = []
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    fs = glob.glob(f"gpurun_out/pmc_{pm}/{c}/**/*counter_collection.csv", recursive=True)
    agg = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "conv3x3" in r["Kernel_Name"] and r["Counter_Name"] == c:
                agg.append(float(r["Counter_Value"]))
    agg = agg[-4:]
    vals[c] = sum(agg) / len(agg)
    for f in glob.glob(f"gpurun_out/pmc_{pm}/{c}/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "conv3x3" in r["Kernel_Name"]]
        if c == "GRBM_GUI_ACTIVE":
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-4:]
fetch, write = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
alg = 2 * 32 * 128 * 128 * 128 * 2
us = sum(dur) / len(dur)
cycles = vals["GRBM_GUI_ACTIVE"] / 8
out = {
    "kernel": "3x3 128->128 forward (the persistent 8-wave kernel the library routes it to: conv3x3_cs_kernel from round 5) @ [32,128,128,128], affine + ReLU on load, batch-norm statistics epilogue",
    "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM)",
    "hbm_bytes_per_launch": int(fetch + write), "algorithmic_bytes_per_launch": alg,
    "hbm_over_algorithmic": round((fetch + write) / alg, 3),
    "SQ_VALU_MFMA_BUSY_CYCLES": vals["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": vals["GRBM_GUI_ACTIVE"],
    "kernel_us_under_the_counter_pass": round(us, 1),
    "mfma_busy_frac": round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles), 3),
    "effective_clock_GHz": round(cycles / us / 1e3, 2),
    "collected_with": f"tools/pmc_passes.sh {pm} 'FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE' tools/one_conv.py fwd 128 128 128 3 "
                      "(rocprofv3 --pmc, one counter per pass, --kernel-trace; mean of the last 4 of 5 launches)",
}
json.dump(out, open(f"profiles/{tag}_dominant_kernel_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
