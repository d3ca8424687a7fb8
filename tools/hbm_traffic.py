"""Turn the two PMC summaries (tools/rocpd_summary.py --counters on a FETCH_SIZE pass and a WRITE_SIZE pass) into
profiles/hbm_traffic.json, the per-stream HBM traffic bench.py reports as roofline.traffic.
usage: python tools/hbm_traffic.py <fetch.csv> <write.csv> <batch> <out.json> "<profiled command>" """
import csv, json, sys


def per_launch(path, counter, kernel_substr):
    vals = []
    for row in csv.reader(l for l in open(path) if not l.startswith("#")):
        if len(row) == 6 and row[1] == counter and kernel_substr in row[0]:
            vals.append(float(row[5]))
    vals = vals[1:] if len(vals) > 1 else vals      # first launch of the process = warm-up (cold caches, page faults)
    return sum(vals) / max(len(vals), 1), len(vals)


fetch_csv, write_csv, batch, out, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
res = {"note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) on `{cmd}`, MI355X. Counters are in KiB. "
               "gfx950 correction from MI355X_MICROARCH.md (HBM section): FETCH_SIZE reads half the bytes of wide coalesced reads -> doubled; "
               "WRITE_SIZE uncalibrated, taken as is.", "batch": batch}
for kern, key in (("align_fused", "align_fused_kernel"), ("pose_opt", "pose_opt_kernel")):
    f, n = per_launch(fetch_csv, "FETCH_SIZE", kern)
    w, _ = per_launch(write_csv, "WRITE_SIZE", kern)
    res[f"{key}_launches_measured"] = n
    res[f"{key}_FETCH_SIZE_KiB_per_launch"] = f
    res[f"{key}_WRITE_SIZE_KiB_per_launch"] = w
    res[f"{key}_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    res[f"{key}_bytes_per_stream"] = (2.0 * f + w) * 1024.0 / batch
res["launches_measured"] = res["align_fused_kernel_launches_measured"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
