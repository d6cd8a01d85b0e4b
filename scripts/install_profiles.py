"""Copies the summaries scripts/collect_profiles.sh produced (gpurun_out/<dir>) into profiles/ under the round's names and
regenerates profiles/traffic.json.  usage: python3 scripts/install_profiles.py gpurun_out/<dir> r02

traffic.json: HBM-side bytes per launch from the PMC counters, FETCH_SIZE x 2 + WRITE_SIZE (KiB).  The x2 is the gfx950
correction MI355X_MICROARCH.md prescribes, confirmed here on known byte counts for 16-, 4- and 1-byte-per-lane reads
(tools/fetch_calib.hip, <tag>_fetch_calibration.txt): the counter tallies 128-byte requests at 64 bytes.  The file records
the sha256 of the kernel sources it was measured on; bench.py reports `roofline.traffic` only when that matches."""
import hashlib
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
for a, b in [("bench_default.json", "bench_default.json"), ("stats_default.csv", "stats_default_kernel_stats.csv"),
             ("stats_1slot.csv", "stats_1slot_kernel_stats.csv"), ("fetch.txt", "pmc_fetch_size_summary.txt"),
             ("write.txt", "pmc_write_size_summary.txt"), ("sq.txt", "pmc_sq_counters.txt"), ("calib.txt", "fetch_calibration.txt"),
             ("bow_rate.json", "bow_rate.json"), ("lf_rate.json", "lf_rate.json"), ("n1_rate.json", "n1_rate.json"),
             ("latency.json", "latency.json"), ("latency_nograph.json", "latency_nograph.json"), ("latency_hostselect.json", "latency_hostselect.json"),
             ("fp4_probe.txt", "fp4_probe.txt"), ("valu_rates.txt", "valu_rates.txt"), ("lat_probe.txt", "lat_probe.txt"),
             ("latency_trace.txt", "latency_trace.txt")]:
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(P, "%s_%s" % (tag, b)))


def kib(path):
    d, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.split()[0]
        else:
            d[cur] = float(line.split()[2])
    return d


bench = json.load(open(os.path.join(P, tag + "_bench_default.json")))
NIMG = 128   # the counter runs of scripts/collect_profiles.sh launch 32 four-camera rig frames (--slots 1 --frames 32), whatever the bench line's batch is
f, w = kib(os.path.join(P, tag + "_pmc_fetch_size_summary.txt")), kib(os.path.join(P, tag + "_pmc_write_size_summary.txt"))
sha = hashlib.sha256(open(os.path.join(ROOT, "mc-slam_amd", "csrc", "mcorb_kernels.hip"), "rb").read()).hexdigest()[:16]
out = {"_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 "
                  "--warmup 1 --repeats 1 --no-cpu --no-latency --no-staging --slots 1 --frames 32 --iso-jobs 1 (%d images per launch), "
                  "MI355X, ROCm 7.2; per-kernel means in profiles/%s_pmc_*_size_summary.txt" % (NIMG, tag),
       "_correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: FETCH_SIZE reads exactly 1/2 of a known byte count for 16-, 4- "
                      "and 1-byte-per-lane streaming reads on this gfx950 (profiles/%s_fetch_calibration.txt), WRITE_SIZE is exact" % tag,
       "kernels_sha256_16": sha, "config": bench["config"]["name"], "round": tag}
# vector instructions per launch (SQ_INSTS_VALU), for the instruction-issue bound bench.py reports beside the HBM one
valu, mfma, cur = {}, {}, None
sq = os.path.join(P, tag + "_pmc_sq_counters.txt")
if os.path.exists(sq):
    for line in open(sq):
        if not line.startswith(" "):
            cur = line.split()[0]
        elif line.split()[0] == "SQ_INSTS_VALU" and cur not in valu:
            valu[cur] = float(line.split()[2])
        elif line.split()[0] == "SQ_INSTS_MFMA" and cur not in mfma:
            mfma[cur] = float(line.split()[2])
for k in sorted(set(f) & set(w)):
    if not k.startswith("k_"):
        continue
    mult = 7 if k == "k_resize" else 1   # 7 level launches per image batch; the summary holds the per-launch mean
    out[k] = {"bytes_per_image": int((2 * f[k] + w[k]) * 1024 * mult / NIMG), "fetch_size_kib_per_launch_raw": f[k],
              "write_size_kib_per_launch": w[k], "images_per_launch": NIMG}
    if k in valu:
        out[k]["valu_insts_per_launch"] = int(valu[k] * mult)
    if mfma.get(k):
        out[k]["mfma_insts_per_launch"] = int(mfma[k] * mult)
    if mult > 1:
        out[k]["note"] = "sum of the 7 level launches"
json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print("installed", tag, "kernels sha", sha)
