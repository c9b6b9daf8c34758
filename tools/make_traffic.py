"""profiles/traffic.json from a tools/pmc_summary.py summary: HBM bytes per launch and kernel.

traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both counters in KiB, and on gfx950
FETCH_SIZE counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section).
Usage: python tools/make_traffic.py SUMMARY.json WORKLOAD OUT.json
"""
import json
import sys

STAGE_OF = {"render_bwd_kernel": "render_bwd", "render_fwd_kernel": "render_fwd", "segment_reduce_kernel": "segment_reduce",
            "gaussian_bwd_kernel": "gaussian_bwd", "preprocess_kernel": "preprocess", "emit_kernel": "emit",
            "rs_scatter_kernel": "sort_scatter", "rs_hist_kernel": "sort_hist", "l1_partial_kernel": "l1_loss"}


def main():
    summary, workload, out = sys.argv[1:4]
    s = json.load(open(summary))
    res, raw = {}, {}
    for k, stage in STAGE_OF.items():
        c = s.get(k)
        if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        res[stage] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        raw[stage] = {"FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"]}
    try:
        doc = json.load(open(out))
    except Exception:
        doc = {}
    doc["note"] = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_pmc.sh); "
                   "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B requests at 64 B "
                   "(MI355X_MICROARCH.md, HBM section); sort kernels: mean over all launches of the frame")
    doc[workload] = res
    doc["raw_" + workload] = raw
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
