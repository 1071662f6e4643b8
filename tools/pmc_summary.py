"""Mean per-dispatch counter values of dev::k_primary from rocprofv3 --pmc csv output."""
import collections, csv, glob, os, sys
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_primary<" in r["Kernel_Name"] or r["Kernel_Name"].startswith("dev::k_primary("):   # the main kernel, not k_primary_exact
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    group = os.path.relpath(f, root).split(os.sep)[0]
    for k, v in acc.items():
        print("%-6s %-24s n=%-3d mean=%.6g" % (group, k, len(v), sum(v) / len(v)))
