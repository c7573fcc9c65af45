"""Sums rocprofv3 --pmc counter_collection.csv files (any SQ_* counters, one or more passes of
`bench.py --steps S --warmup W`) per kernel and counter, per bench step:
python3 tools/summarize_sq.py <steps_total> <out.json> <csv> [<csv> ...]"""
import collections, csv, json, sys

NAMES = {"frame_yin_kernel": "frame", "pyin_obs_kernel": "pyin_obs", "viterbi_band_kernel": "viterbi", "viterbi_band_dense_kernel": "viterbi",
         "viterbi_band_split_kernel": "viterbi", "viterbi_kernel": "viterbi", "db_rake_kernel": "db_rake"}


def main(n_steps, out, paths):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in paths:
        for r in csv.DictReader(open(path)):
            key = next((v for k, v in NAMES.items() if k in r["Kernel_Name"]), None)
            if key is not None:
                per[key][r["Counter_Name"]] += float(r["Counter_Value"]) / n_steps
    res = {"units": "per bench step, summed over the step's launches (SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES tick in quad-cycles, MI355X_MICROARCH.md)",
           "per_step": {k: {c: int(v) for c, v in sorted(d.items())} for k, d in per.items()}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(int(sys.argv[1]), sys.argv[2], sys.argv[3:])
