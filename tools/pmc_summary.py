"""Summarise rocprofv3 --pmc output: mean counter value per launch for every kernel.
    python tools/pmc_summary.py OUT_DIR [OUT_DIR ...] > summary.json
Each OUT_DIR is one `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d OUT_DIR -- ...` pass (counters
are collected in separate passes, see /opt/skills/guides/MI355X_MICROARCH.md); the `*_counter_collection.csv` files are
merged by kernel name (template arguments kept, parameter list dropped)."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geometric_adv_amd", "csrc")
HASHED = ("encoder.hip", "encoder_x3.hip", "encoder_x3.h", "mfma_tile.h")      # bench.py drops counters taken at other sources of the dominant kernel
                                             # (--hash a,b,c selects other files: the Chamfer summary hashes its own)


def main(dirs):
    global HASHED
    if dirs and dirs[0] == "--hash":
        HASHED = tuple(dirs[1].split(","))
        dirs = dirs[2:]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(dict)
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                per_dispatch[(r["Dispatch_Id"], name)][r["Counter_Name"]] = per_dispatch[(r["Dispatch_Id"], name)].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                if "Start_Timestamp" in r and r["Start_Timestamp"]:
                    per_dispatch[(r["Dispatch_Id"], name)]["__dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
            for (_, name), cs in per_dispatch.items():
                for c, v in cs.items():
                    (dur[name] if c == "__dur" else acc[name][c]).append(v)
    out = {}
    for name, cs in acc.items():
        out[name] = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()}
        if dur[name]:
            out[name]["avg_us_profiled"] = sum(dur[name]) / len(dur[name])
    out["_source_sha1"] = {f: hashlib.sha1(open(os.path.join(CSRC, f), "rb").read()).hexdigest()[:12] for f in HASHED}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
