"""Mean counter values of one kernel from rocprofv3 --pmc CSVs: python tools/pmc_summary.py <substring of kernel name> <dir> [<dir> ...]"""
import collections, csv, glob, sys
name = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v = v[-4:] if len(v) > 4 else v
            print(f"{k:36s} {sum(v) / len(v):16.0f}   (n={len(v)})")
