"""mean counter value per kernel from rocprofv3 --pmc CSV directories (first dispatch of each kernel dropped: warm-up)"""
import csv, glob, json, sys, collections
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_") or "k_decode" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:34], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            v = v[1:] if len(v) > 1 else v
            out["%s|%s" % (k, c)] = sum(v) / len(v)
print(json.dumps(out, indent=1))
