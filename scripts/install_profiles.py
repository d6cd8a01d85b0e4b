"""Copies the summaries scripts/collect_profiles.sh produced (gpurun_out/<dir>) into profiles/ under the round's names and
regenerates profiles/traffic.json.  usage: python3 scripts/install_profiles.py gpurun_out/<dir> r01"""
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
for a, b in [("bench_default.json", "bench_default.json"), ("p_def/d_kernel_stats.csv", "stats_default_kernel_stats.csv"),
             ("p_1s/s_kernel_stats.csv", "stats_1slot_kernel_stats.csv"), ("fetch.txt", "pmc_fetch_size_summary.txt"),
             ("write.txt", "pmc_write_size_summary.txt"), ("upload_rate.txt", "upload_rate.txt"), ("latency.json", "latency.json"),
             ("bow_rate.json", "bow_rate.json")]:
    shutil.copy(os.path.join(src, a), os.path.join(P, "%s_%s" % (tag, b)))


def kib(path):
    d, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.split()[0]
        else:
            d[cur] = float(line.split()[2])
    return d


NIMG = json.load(open(os.path.join(P, tag + "_bench_default.json")))["roofline"]["images_per_launch"]   # the one-slot runs use the same launch size
f, w = kib(os.path.join(P, tag + "_pmc_fetch_size_summary.txt")), kib(os.path.join(P, tag + "_pmc_write_size_summary.txt"))
old = json.load(open(os.path.join(P, "traffic.json")))
out = {"_source": old["_source"]}
for k in ("k_fast_cells", "k_blur", "k_knn2", "k_describe", "k_compact", "k_knn2_finalize"):
    out[k] = {"bytes_per_image": int((f[k] + w[k]) * 1024 / NIMG), "fetch_kib_per_launch": f[k], "write_kib_per_launch": w[k], "images_per_launch": NIMG}
out["k_resize"] = {"bytes_per_image": int((f["k_resize"] + w["k_resize"]) * 1024 * 7 / NIMG), "note": "sum of the 7 level launches",
                   "fetch_kib_per_launch": f["k_resize"], "write_kib_per_launch": w["k_resize"]}
json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print("installed", tag)
