"""Merge gpurun_out/prof_<tag>/pmc.json records into profiles/pmc_latest.json (one record per
bench.py configuration: streams, geometry, frames per step, input) and copy each run's
summary / kernel-stats CSV into profiles/ under a per-round name.
    merge_pmc.py <round tag, e.g. r02> <prof dir> [<prof dir> ...]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def main(rnd, dirs):
    path = os.path.join(ROOT, "pmc_latest.json")
    try:
        recs = json.load(open(path))
    except Exception:
        recs = []
    for d in dirs:
        try:
            rec = json.load(open(os.path.join(d, "pmc.json")))
        except Exception as e:
            print("skip %s: %s" % (d, e))
            continue
        rec["source"] = "profiles/%s_rocprof_summary_%s.txt" % (rnd, os.path.basename(d).replace("prof_", ""))
        recs = [r for r in recs if r.get("config") != rec["config"]] + [rec]
        tag = os.path.basename(d).replace("prof_", "")
        shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "%s_rocprof_summary_%s.txt" % (rnd, tag)))
        for f in os.listdir(os.path.join(d, "keep")) if os.path.isdir(os.path.join(d, "keep")) else []:
            shutil.copy(os.path.join(d, "keep", f), os.path.join(ROOT, "%s_kernel_stats_%s.csv" % (rnd, tag)))
        bl = os.path.join(d, "bench_line.json")
        if os.path.exists(bl) and os.path.getsize(bl):
            shutil.copy(bl, os.path.join(ROOT, "%s_bench_traced_%s.json" % (rnd, tag)))
    json.dump(recs, open(path, "w"), indent=1)
    print("pmc_latest.json:", [r["config"] for r in recs])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
