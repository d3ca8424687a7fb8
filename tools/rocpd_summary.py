"""Summarise a rocprofv3 rocpd database (…_results.db) into a small CSV for profiles/.
usage: python tools/rocpd_summary.py <results.db> <out.csv> "<command line that was profiled>" """
import sqlite3
import sys


def main(dbs, out, cmd):
    """dbs: one rocpd database per traced PROCESS, comma separated (bench.py's latency leg runs adapter_driver as a child process,
    which rocprofv3 traces into its own database): every process gets its own block, the one with the most kernel time first."""
    blocks = []
    for db in dbs.split(","):
        c = sqlite3.connect(db)
        q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
             "where name like '%plsvo%' group by name order by sum(duration) desc")
        totals = list(c.execute(q))
        q = ("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, duration "
             "from kernels where name like '%align_fused%' or name like '%align_level%' or name like '%pose_opt%' order by start")
        blocks.append((sum(r[2] for r in totals), db, totals, list(c.execute(q))))
    blocks.sort(key=lambda b: -b[0])
    lines = [f"# rocprofv3 --kernel-trace --stats -- {cmd}",
             "# extracted from the rocpd databases (view `kernels`), one block per traced process; durations in microseconds"]
    for k, (tot, db, totals, rows) in enumerate(blocks):
        if not totals:
            continue
        who = "main process (python bench.py)" if k == 0 else "child process (adapter_driver --bench: the drop-in's per-call latency leg)"
        lines += ["", f"## process {k}: {who} -- per-kernel totals (plsvo kernels only; torch kernels generate the synthetic inputs)",
                  "name,calls,total_us,avg_us,min_us,max_us"]
        for r in totals:
            lines.append(f"\"{r[0]}\",{r[1]},{r[2] / 1e3:.1f},{r[3] / 1e3:.1f},{r[4] / 1e3:.1f},{r[5] / 1e3:.1f}")
        if k == 0:
            lines += ["", "## process 0: every align_fused_kernel / pose_opt_kernel dispatch with more than 1024 workgroups, in launch order "
                          "(the small-batch launches of the latency leg are summarised above only)",
                      "name,grid_x,workgroup_x,lds_bytes,arch_vgpr,accum_vgpr,sgpr,scratch,duration_us"]
            for r in rows:
                if r[1] // max(r[2], 1) > 1024:
                    lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]},{r[4]},{r[5]},{r[6]},{r[7]},{r[8] / 1e3:.1f}")
    open(out, "w").write("\n".join(lines) + "\n")




def per_kernel(dbs, out, cmd):
    """one row per (kernel, launch shape) of the plsvo kernels, all traced processes together: call count, time, resources"""
    lines = [f"# rocprofv3 --kernel-trace --stats -- {cmd}", "# extracted from the rocpd databases (view `kernels`); durations in microseconds",
             "name,grid_x,workgroup_x,calls,total_us,avg_us,min_us,max_us,lds_bytes,arch_vgpr,sgpr,scratch"]
    for db in dbs.split(","):
        c = sqlite3.connect(db)
        q = ("select name, grid_x, workgroup_x, count(*), sum(duration), avg(duration), min(duration), max(duration), max(lds_size), "
             "max(vgpr_count), max(sgpr_count), max(scratch_size) from kernels where name like '%plsvo%' "
             "group by name, grid_x, workgroup_x order by sum(duration) desc")
        for r in c.execute(q):
            lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]},{r[4] / 1e3:.1f},{r[5] / 1e3:.1f},{r[6] / 1e3:.1f},{r[7] / 1e3:.1f},{r[8]},{r[9]},{r[10]},{r[11]}")
    open(out, "w").write("\n".join(lines) + "\n")


def counters(dbs, out, cmd, like=None):
    """per-kernel PMC counter totals (rocprofv3 --pmc ... pass) for the plsvo kernels (or kernels matching `like`); dbs comma separated"""
    lines = [f"# rocprofv3 --kernel-trace --pmc <counter> -- {cmd}", "# view counters_collection; one row per dispatch",
             "kernel,counter,dispatch_index,grid,workgroup,value"]
    for db in dbs.split(","):
        _counters_one(db, lines, like)
    open(out, "w").write("\n".join(lines) + "\n")


def _counters_one(db, lines, like):
    c = sqlite3.connect(db)
    where = ("kernel_name like '%plsvo%align_fused%' or kernel_name like '%plsvo%pose_opt%'" if not like
             else " or ".join(f"kernel_name like '{l}'" for l in like.split(",")))
    q = ("select kernel_name, counter_name, dispatch_id, grid_size, workgroup_size, value from counters_collection "
         f"where {where} order by dispatch_id")
    try:
        for r in c.execute(q):
            lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]},{r[4]},{r[5]}")
    except sqlite3.Error:
        pass   # a traced child process without counter rows


if __name__ == "__main__":
    if sys.argv[1] == "--counters":
        counters(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "", sys.argv[5] if len(sys.argv) > 5 else None)
    elif sys.argv[1] == "--per-kernel":
        per_kernel(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    else:
        main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
