"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel (last dispatch of each kernel)."""
import csv, sys, collections, re


def kname(n):
    """mcorb::k_x(args) / void mcorb::k_x<48>(args) -> k_x"""
    return re.sub(r"<.*?>", "", n.split("(")[0]).replace("void ", "").replace("mcorb::", "").strip()

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = kname(r["Kernel_Name"])
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
extra = {}
for r in rows:
    k = kname(r["Kernel_Name"])
    extra[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
for k, d in acc.items():
    print(k, "vgpr/sgpr/lds/grid/wg", extra[k])
    for c, v in sorted(d.items()):
        print("    %-24s mean %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
