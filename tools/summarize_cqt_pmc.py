"""Reads the rocprofv3 --pmc counter_collection.csv of `tools/bench_cqt.py` into profiles/r1_cqt_pmc.json: counters of
the largest launch of each CQT kernel (the 64-clip batch; the 2-clip warm-up launch is dropped), summed over the
dispatch's rows (one per XCD / shader engine)."""
import collections, csv, json, sys


def main(path, out):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        if "aegis::cqt" not in r["Kernel_Name"]:
            continue
        per[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    res = {}
    for name, disp in per.items():
        big = max(disp.values(), key=lambda c: c.get("SQ_INSTS_VALU_MFMA_F32", 0.0))
        res[name] = dict(big)
        res[name]["launches_seen"] = len(disp)
        if big.get("SQ_INSTS_VALU_MFMA_F32"):
            res[name]["mfma_busy_cycles_per_instruction"] = big["SQ_VALU_MFMA_BUSY_CYCLES"] / big["SQ_INSTS_VALU_MFMA_F32"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
