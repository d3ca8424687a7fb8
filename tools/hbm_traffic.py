"""Turn the PMC summaries (tools/rocpd_summary.py --counters on a FETCH_SIZE pass and a WRITE_SIZE pass of bench.py, plus the same
two passes of tools/pmc_calib) into profiles/hbm_traffic.json, the per-stream memory-side traffic bench.py reports as
roofline.traffic (labelled offline, with this file's commit tag).

usage: python tools/hbm_traffic.py <fetch.csv> <write.csv> <batch> <out.json> "<profiled command>" <commit> [<calib_fetch.csv> <calib_write.csv> <calib.json>]

Counters are in KiB.  MI355X_MICROARCH.md (HBM section) says FETCH_SIZE reads exactly half the bytes of a WIDE (16 B per lane)
coalesced stream on gfx950 and calls other access widths and WRITE_SIZE uncalibrated.  The calibration kernels
(tools/pmc_calib.hip) measure the factors in this repository's own access patterns:
  c_wide   = known bytes / FETCH_SIZE for a 16-B-per-lane streaming read   (the ref/dx/dy cache reads of align_fused_kernel)
  c_write  = known bytes / WRITE_SIZE for a 16-B-per-lane streaming write
  gather   : FETCH_SIZE against the DISTINCT 32 / 64 / 128-byte blocks the 3-rows x 2-dwords image gather touches (one
             workgroup per image, as in align_fused_kernel).  Measured: the counter is NOT below the distinct-block bytes for
             this pattern (it reads ~1.6x the distinct 128-B lines), so the gathered share of the fetches gets no upward
             correction (factor 1); only the wide share is doubled.
Three figures are written per kernel: uncorrected (FETCH + WRITE), upper bound (2 x FETCH + WRITE: every fetch wide) and the
estimate bench.py reports, c_fetch x FETCH + c_write x WRITE with c_fetch = w_wide x c_wide + (1 - w_wide) x 1, where w_wide is the
wide share of the kernel's own requests (64 + 24 B of dense loads against 40 B of gathered dwords per patch-iteration; rounds 1-3: 192 + 24 against 48)."""
import csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha     # hash of the sources align_fused_kernel is built from: bench.py uses the traffic only while it matches


def per_launch(path, counter, kernel_substr):
    vals = []
    for row in csv.reader(l for l in open(path) if not l.startswith("#")):
        if len(row) == 6 and row[1] == counter and kernel_substr in row[0]:
            vals.append(float(row[5]))
    vals = vals[1:] if len(vals) > 1 else vals      # first launch of the process = warm-up (cold caches, page faults)
    return sum(vals) / max(len(vals), 1), len(vals)


fetch_csv, write_csv, batch, out, cmd, commit = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
res = {"note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) on `{cmd}`, MI355X. Counters are in KiB.",
       "batch": batch, "measured_at": commit, "kernel_source_sha": kernel_source_sha()}
c_wide, c_gather, c_write, calib = 2.0, 1.0, 1.0, None
if len(sys.argv) > 9:
    cf, cw, cj = sys.argv[7], sys.argv[8], json.load(open(sys.argv[9]))
    f_stream, _ = per_launch(cf, "FETCH_SIZE", "calib_stream_read")
    f_gather, _ = per_launch(cf, "FETCH_SIZE", "calib_gather_dword")
    w_stream, _ = per_launch(cw, "WRITE_SIZE", "calib_stream_write")
    calib = {"known": cj, "FETCH_SIZE_KiB_stream_read": f_stream, "FETCH_SIZE_KiB_gather": f_gather, "WRITE_SIZE_KiB_stream_write": w_stream}
    if f_stream > 0:
        c_wide = cj["stream_read_bytes"] / (f_stream * 1024.0)
    c_gather = 1.0
    if f_gather > 0:
        for g in (32, 64, 128):
            calib[f"gather_distinct_{g}B_block_bytes_over_FETCH_SIZE"] = cj[f"gather_distinct_{g}B_blocks_bytes"] / (f_gather * 1024.0)
    if w_stream > 0:
        c_write = cj["stream_write_bytes"] / (w_stream * 1024.0)
    calib.update({"c_wide": c_wide, "c_gather_used": c_gather, "c_write": c_write})
    res["calibration"] = calib
else:
    res["note"] += " No calibration pass given: FETCH_SIZE doubled (the guide's wide-read figure) for every fetch, WRITE_SIZE as is."
w_wide = (64.0 + 24.0) / (64.0 + 24.0 + 40.0)     # request mix of align_fused_kernel per patch-iteration (bench.py OWN_BYTES_PER_PATCH_ITER): 64-B record + 3-D point wide, 5 rows x 2 dwords gathered
c_fetch = w_wide * c_wide + (1.0 - w_wide) * c_gather
res["fetch_correction_used"] = c_fetch
res["write_correction_used"] = c_write
for kern, key in (("align_fused", "align_fused_kernel"), ("pose_opt", "pose_opt_kernel")):
    f, n = per_launch(fetch_csv, "FETCH_SIZE", kern)
    w, _ = per_launch(write_csv, "WRITE_SIZE", kern)
    res[f"{key}_launches_measured"] = n
    res[f"{key}_FETCH_SIZE_KiB_per_launch"] = f
    res[f"{key}_WRITE_SIZE_KiB_per_launch"] = w
    res[f"{key}_bytes_per_launch"] = (c_fetch * f + c_write * w) * 1024.0
    res[f"{key}_bytes_per_stream"] = (c_fetch * f + c_write * w) * 1024.0 / batch
    res[f"{key}_bytes_per_stream_uncorrected"] = (f + w) * 1024.0 / batch
    res[f"{key}_bytes_per_stream_upper_bound"] = (2.0 * f + c_write * w) * 1024.0 / batch
res["launches_measured"] = res["align_fused_kernel_launches_measured"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
