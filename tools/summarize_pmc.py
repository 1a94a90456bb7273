"""Summarise rocprofv3 --pmc CSVs into per-kernel per-launch numbers (profiles/*.json).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced streaming reads, so
it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores (our epilogues store 8 B per
lane: uncalibrated width, taken as is). Collected in separate passes (TCC slots).
usage: summarize_pmc.py OUT.json name=path.csv [name=path.csv ...]
"""
import collections
import csv
import json
import sys


def load(path):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return by, dur


def main():
    out_path, specs = sys.argv[1], sys.argv[2:]
    res = collections.defaultdict(dict)
    for spec in specs:
        _, path = spec.split("=", 1)
        by, dur = load(path)
        for k, ctrs in by.items():
            if not any(t in k for t in ("gemm", "attn", "rmsnorm", "rope", "lru_", "item_", "topk", "head", "embed", "em_", "cand_",
                                        "bound_", "rms_rstd")):
                continue
            e = res[k]
            e["launches_profiled"] = max(e.get("launches_profiled", 0), len(dur[k]))
            e.setdefault("avg_duration_us", sum(dur[k].values()) / len(dur[k]) / 1e3)
            for c, v in ctrs.items():
                e[c + "_per_launch"] = sum(v) / len(v)
    for k, e in res.items():
        if "FETCH_SIZE_per_launch" in e:
            e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE_per_launch"] * 1024 * 2
        if "WRITE_SIZE_per_launch" in e:
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE_per_launch"] * 1024
        if "hbm_read_bytes_per_launch" in e and "hbm_write_bytes_per_launch" in e:
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        if "TCC_HIT_sum_per_launch" in e and "TCC_MISS_sum_per_launch" in e:
            tot = e["TCC_HIT_sum_per_launch"] + e["TCC_MISS_sum_per_launch"]
            e["l2_hit_rate"] = e["TCC_HIT_sum_per_launch"] / tot if tot else None
        if "TCC_EA0_RDREQ_sum_per_launch" in e and "TCC_EA0_RDREQ_DRAM_sum_per_launch" in e and e["TCC_EA0_RDREQ_sum_per_launch"]:
            # requests the L2 sends out that are destined for DRAM (the memory-side Infinity Cache sits in front of DRAM and
            # has no counter of its own in rocprofv3 -L on gfx950: this ratio does NOT separate MALL hits from HBM reads)
            e["ea_rdreq_dram_fraction"] = e["TCC_EA0_RDREQ_DRAM_sum_per_launch"] / e["TCC_EA0_RDREQ_sum_per_launch"]
        if "GRBM_GUI_ACTIVE_per_launch" in e and "SQ_VALU_MFMA_BUSY_CYCLES_per_launch" in e:
            g = e["GRBM_GUI_ACTIVE_per_launch"] / 8
            e["mfma_busy_fraction"] = e["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / (1024 * g)
            e["effective_clock_ghz"] = g / (e["avg_duration_us"] * 1e-6) / 1e9
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    for k, e in sorted(res.items()):
        print(k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in e.items()
                  if x in ("avg_duration_us", "hbm_bytes_per_launch", "mfma_busy_fraction", "effective_clock_ghz", "l2_hit_rate",
                           "ea_rdreq_dram_fraction")})


if __name__ == "__main__":
    main()
