"""Summarises rocprofv3 --pmc counter_collection.csv files (FETCH_SIZE / WRITE_SIZE passes of
`bench.py --steps S --warmup W`) into profiles/r1_pmc_hbm.json: KB per bench step and kernel."""
import collections, csv, json, sys

NAMES = {"frame_yin_kernel": "frame", "pyin_obs_kernel": "pyin_obs",
         "viterbi_band_kernel": "viterbi", "viterbi_band_dense_kernel": "viterbi", "viterbi_kernel": "viterbi", "decode_kernel": "finalize",
         "db_rake_kernel": "finalize", "rake_runs_kernel": "finalize"}


def main(fetch_csv, write_csv, n_steps_total, out):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    for counter, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter or "aegis::" not in r["Kernel_Name"]:
                continue
            key = next((v for k, v in NAMES.items() if k in r["Kernel_Name"]), None)
            if key is None:
                continue
            per[key][counter + "_KB"] += float(r["Counter_Value"]) / n_steps_total
            if counter == "FETCH_SIZE":
                launches[key] += 1
    res = {"units": "KB per bench step (sum over the kernel's launches in one step); rocprofv3 --pmc, separate passes",
           "note": "gfx950: FETCH_SIZE counts 64 B per 128 B request on wide coalesced reads -> double it (MI355X_MICROARCH.md, HBM)",
           "steps_profiled": n_steps_total,
           "schedule": "AEGIS_VITERBI_PERSISTENT=0: counter collection serialises kernels, so the Viterbi runs one launch per time chunk",
           "per_step": {k: dict(v, launches_per_step=launches[k] / n_steps_total) for k, v in per.items()}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
