#!/usr/bin/env python3
"""DESIGN.md = the parts under docs/design/ in order, with the @PLACEHOLDERS@ of the measurement section filled from a bench line
and profiles/traffic.json, so that the numbers in the text are the numbers of one committed run.

    python3 scripts/assemble_design.py [profiles/r04_bench_default.json]
"""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_bench_default.json")
if os.path.isdir(src):
    src = os.path.join(src, "bench_default.json")
d = json.loads([l for l in open(src).read().strip().splitlines() if l.startswith("{")][-1])
tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
ipl = d["roofline"]["images_per_launch"]
iso = {k: v * 128.0 / ipl for k, v in d["kernel_us_per_launch_isolated"].items()}   # per 128 images (192 pairs)
alg = {"k_fast_cells": 381.9, "k_resize": 606.8, "k_describe_fused": 287.8, "k_knn2": 30.8, "k_compact": 41.1}
def traffic(k):
    return tj[k]["bytes_per_image"] * 128 / 1e6 if k != "k_knn2" else tj[k]["bytes_per_image"] * 128 / 1e6
lat = d["single_frame_latency_ms"]
R = {
 "VALUE": "%d" % round(d["value"]), "MSF": "%.4f" % d["ms_per_frame"], "VMIN": "%d" % round(d["value_min"]), "VMAX": "%d" % round(d["value_max"]),
 "VHC": "%d" % round(d["value_host_cores"]["value"]), "VHCF": "%.2f" % (d["value_host_cores"]["value"] / d["value"]),
 "VST": "%d" % round(d["value_with_staging"]), "PCIE": "%.1f" % d["pcie_gbs"], "VU8": "%d" % round(d["value_with_upload_u8"]),
 "VFD": "%d" % round(d["value_force_dist"]["value"]), "VC2": "%d" % round(d["configs2_1080p8"]["value"]),
 "CPU4": "%.1f" % d["cpu_baseline"]["value"], "CPU1": "%.1f" % d["cpu_baseline"]["value_1thread"],
 "SUP": "%.3f" % lat["separated"]["upload_incl_dma_ms"], "SEM": "%.3f" % lat["separated"]["extract_match_ms"],
 "LUP": "%.3f" % lat["upload_ms"], "LEM": "%.3f" % lat["extract_match_ms"], "LRB": "%.3f" % lat["readback_ms"], "LTOT": "%.3f" % lat["total_ms"],
 "WPA": "%.2f" % (d["roofline"]["whole_path"]["achieved"] / 1000), "WPF": "%.3f" % d["roofline"]["whole_path"]["frac"], "WPB": "%.3f" % d["roofline"]["whole_path"]["frac_this_build"],
 "AVG_FAST": "%.0f" % (d["roofline"]["avg_launch_us"] * 128.0 / ipl), "ACH": "%.0f" % d["roofline"]["achieved"], "FRAC": "%.3f" % d["roofline"]["frac"],
 "DRATE": "%.1f" % (770e6 / (iso["k_knn2"] * 1e-6) / 1e12),
}
short = {"k_fast_cells": "FAST", "k_resize": "RES", "k_describe_fused": "DESC", "k_knn2": "KNN", "k_compact": "COMP"}
for k, s in short.items():
    R["K_" + s] = "%.0f" % iso[k]
    R["F_" + s] = "%.3f" % (alg[k] * 1e6 / (iso[k] * 1e-6) / 8e12)
    t = traffic(k)
    R["T_" + s] = "%.1f" % t
    R["R_" + s] = "%.2f" % (t / alg[k])
sel = iso.get("k_select+k_assemble", 42.0)
R["K_SEL"] = "%.0f" % sel
ksum = sum(iso[k] for k in short) + sel + 12
R["KSUM"] = "%.0f" % ksum
R["KIDEAL"] = "%.1f" % (32 / (ksum * 1e-6) / 1000)
R["PIPEF"] = "%.0f" % (100 * d["value"] / (32 / (ksum * 1e-6)))
parts = sorted(glob.glob(os.path.join(ROOT, "docs", "design", "*.md")))
text = "".join(open(p).read() + ("" if open(p).read().endswith("\n\n") else "\n") for p in parts)
for k, v in R.items():
    text = text.replace("@" + k + "@", v)
left = re.findall(r"@[A-Z0-9_]+@", text)
assert not left, left
open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
print("DESIGN.md written,", len(text.splitlines()), "lines;", R)
