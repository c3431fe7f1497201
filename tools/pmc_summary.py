"""Summarise rocprofv3 --pmc passes (csv output) per kernel: HBM bytes per launch and MFMA utilisation.

usage: pmc_summary.py <dir_fetch> <dir_write> <dir_sq> <out_summary.txt> <traffic.json> <key> <commit> <gram_kernel_prefix> <proj_kernel_prefix>
Conventions (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 a wide
coalesced read is counted as 64 B per 128-B request, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Values are those
of the LARGEST launch of each kernel.  traffic.json is keyed "<workload>/<dtype>/<route>" and records the commit the
passes ran on; bench.py only reports `roofline.traffic` for a key it finds there and names the file as the source."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def load(d):
    out = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> values per dispatch
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("ffvd::", "")
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


fetch, write, sq = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
key, commit, gram_prefix, proj_prefix = sys.argv[6], sys.argv[7], sys.argv[8], sys.argv[9]
lines = ["rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py ... (%s, commit %s); separate passes "
         "for FETCH_SIZE, WRITE_SIZE and the SQ/GRBM set" % (key, commit),
         "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (rocprofv3 reports KB; gfx950 FETCH_SIZE counts 64 B per 128-B request "
         "for wide coalesced reads)",
         "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); values are those of the LARGEST "
         "launch of each kernel", ""]
traffic = {}
for k in sorted(set(fetch) | set(write)):
    fmax = max(fetch.get(k, {}).get("FETCH_SIZE", [0.0]))
    wmax = max(write.get(k, {}).get("WRITE_SIZE", [0.0]))
    b = (2 * fmax + wmax) * 1024
    s = "%-44s FETCH_SIZE_max=%11.0f WRITE_SIZE_max=%11.0f hbm_bytes_per_launch=%.4e" % (k[:44], fmax, wmax, b)
    c = sq.get(k, {})
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        i = max(range(len(c["GRBM_GUI_ACTIVE"])), key=lambda j: c["GRBM_GUI_ACTIVE"][j])
        busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"][i], c["GRBM_GUI_ACTIVE"][i]
        if busy > 0 and act > 0:
            s += " mfma_util=%.3f" % (busy / (act / 8 * 1024))
            if c.get("SQ_WAIT_INST_ANY") and c.get("SQ_WAVE_CYCLES") and sum(c["SQ_WAVE_CYCLES"]) > 0:
                s += " wait_any_frac=%.3f" % (sum(c["SQ_WAIT_INST_ANY"]) / sum(c["SQ_WAVE_CYCLES"]))   # all launches
    lines.append(s)
    traffic[k] = b
open(sys.argv[4], "w").write("\n".join(lines) + "\n")
gram = max((v for k, v in traffic.items() if k.startswith(gram_prefix)), default=None)
proj = max((v for k, v in traffic.items() if k.startswith(proj_prefix)), default=None)
tj = {}
if os.path.exists(sys.argv[5]):
    try:
        tj = json.load(open(sys.argv[5]))
    except Exception:
        tj = {}
tj = {k: v for k, v in tj.items() if isinstance(v, dict)}
tj[key] = {"gram_H": gram, "project_F": proj, "_commit": commit,
           "_note": "HBM bytes per launch (largest launch) = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes"}
json.dump(tj, open(sys.argv[5], "w"), indent=1)
print("\n".join(lines))
