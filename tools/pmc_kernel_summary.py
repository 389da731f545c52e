"""Average per-launch values of rocprofv3 --pmc counters for one kernel.  usage: pmc_kernel_summary.py <counter_collection.csv> <kernel substring> <out.json>"""
import csv, hashlib, json, os, sys
from collections import defaultdict


def main():
    path, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]
    tot, cnt = defaultdict(float), defaultdict(int)
    res = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
                for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                    if k in r and k not in res:
                        res[k] = r[k]
    doc = {"kernel": kernel, "launches": max(cnt.values()) if cnt else 0, "per_launch_avg": {k: tot[k] / cnt[k] for k in tot}, "dispatch_info": res,
           "note": "SQ_* counters are summed over all waves of a launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_BUSY_CYCLES count quad-cycles (MI355X_MICROARCH.md)",
           "fnn_hip_sha256": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastneighbornet_amd", "csrc",
                                                              "fnn_hip.hip"), "rb").read()).hexdigest()}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
