"""Per-frame VALU instruction counts by kernel family and op class from the PMC passes of tools/valu_frame.sh:
   python tools/summarize_valu.py <raw dir> <images in each pass> <out dir>"""
import collections
import csv
import glob
import json
import os
import re
import sys

raw, n_img, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def family(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"siftk::", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"<.*$", "", name)
    return name.strip()


tot = collections.defaultdict(lambda: collections.defaultdict(float))      # family -> counter -> sum over the run
launches = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(raw, "g*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        tot[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[fam].add((os.path.dirname(f), r["Dispatch_Id"]))
fams = sorted(tot, key=lambda k: -tot[k].get("SQ_INSTS_VALU", 0))
classes = ["ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "INT32", "INT64", "CVT"]
res = {"images_per_pass": n_img, "families": {}, "note": "wave-level instruction counts per headline frame (4096^2 white noise, 3 octaves), "
       "rocprofv3 --pmc, every launch of the run summed and divided by the images of the run"}
lines = ["%-28s %10s %8s | %s | %8s %8s" % ("kernel family", "VALU/frame", "SALU", " ".join("%9s" % c for c in classes), "other", "LDS")]
gsum = collections.defaultdict(float)
for fam in fams:
    c = tot[fam]
    per = {k: v / n_img for k, v in c.items()}
    valu = per.get("SQ_INSTS_VALU", 0.0)
    cls = {k: per.get("SQ_INSTS_VALU_" + k, 0.0) for k in classes}
    other = valu - sum(cls.values())
    res["families"][fam] = {"valu": valu, "salu": per.get("SQ_INSTS_SALU", 0.0), "lds": per.get("SQ_INSTS_LDS", 0.0), "classes": cls, "other": other,
                            "active_inst_valu": per.get("SQ_ACTIVE_INST_VALU", 0.0), "wave_cycles": per.get("SQ_WAVE_CYCLES", 0.0),
                            "wait_any": per.get("SQ_WAIT_ANY", 0.0), "lds_bank_conflict": per.get("SQ_LDS_BANK_CONFLICT", 0.0),
                            "lds_idx_active": per.get("SQ_LDS_IDX_ACTIVE", 0.0)}
    for k, v in cls.items():
        gsum[k] += v
    gsum["valu"] += valu; gsum["other"] += other; gsum["salu"] += per.get("SQ_INSTS_SALU", 0.0); gsum["lds"] += per.get("SQ_INSTS_LDS", 0.0)
    if valu > 1e4:
        lines.append("%-28s %10.3fM %7.2fM | %s | %7.2fM %7.2fM" % (fam[:28], valu / 1e6, per.get("SQ_INSTS_SALU", 0.0) / 1e6,
                     " ".join("%8.2fM" % (cls[k] / 1e6) for k in classes), other / 1e6, per.get("SQ_INSTS_LDS", 0.0) / 1e6))
lines.append("%-28s %10.3fM %7.2fM | %s | %7.2fM %7.2fM" % ("ALL", gsum["valu"] / 1e6, gsum["salu"] / 1e6,
             " ".join("%8.2fM" % (gsum[k] / 1e6) for k in classes), gsum["other"] / 1e6, gsum["lds"] / 1e6))
res["total"] = {"valu": gsum["valu"], "salu": gsum["salu"], "lds": gsum["lds"], "classes": {k: gsum[k] for k in classes}, "other": gsum["other"]}
os.makedirs(out, exist_ok=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sift_pyocl_amd import _lib as _siftlib
res["library_fingerprint"] = _siftlib.source_fingerprint()
json.dump(res, open(os.path.join(out, "valu_frame.json"), "w"), indent=1)
open(os.path.join(out, "valu_frame.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
