"""Per-kernel mean counter values per dispatch from rocprofv3 --pmc csv output (and dispatches per kernel).
usage: python tools/pmc_summary.py <dir> [kernel-name substring ...]   (default: the main primary kernel only, as in rounds 1-2; `all` = every kernel)
Lines: <group> <counter> n=<dispatches> mean=<value> [kernel=<short name>]"""
import collections, csv, glob, os, re, sys
root = sys.argv[1]
want = sys.argv[2:]
def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not want:
            if "k_primary<" in k or k.startswith("dev::k_primary("):   # the main kernel, not k_primary_exact
                acc[("", r["Counter_Name"])].append(float(r["Counter_Value"]))
        elif want == ["all"] or any(w in k for w in want):
            acc[(short(k), r["Counter_Name"])].append(float(r["Counter_Value"]))
    group = os.path.relpath(f, root).split(os.sep)[0]
    for (kn, c), v in sorted(acc.items()):
        print("%-6s %-24s n=%-4d mean=%.6g%s" % (group, c, len(v), sum(v) / len(v), (" kernel=" + kn) if kn else ""))
