#!/bin/bash
# Round-6 counter survey of align_fused_kernel<64> (staged order): what the memory pipeline of a CU is doing while the launch runs.
# Small batch (8192 streams at the one-wave-per-frame shape) so that every pass takes a minute; counters restricted to the kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06d
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $O/avail.txt 2>&1 || rocprofv3 --list-avail > $O/avail.txt 2>&1
wc -l $O/avail.txt
CMD="python $R/bench.py --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
export PLSVO_ALIGN_THREADS=64 PLSVO_BENCH_LAUNCH_ORDER=staged
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "MemUnitBusy MemUnitStalled VALUBusy" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  rm -rf /tmp/pmcs_$i
  T0=$SECONDS
  timeout 400 rocprofv3 --kernel-trace --pmc $SET --kernel-include-regex "align_fused" -d /tmp/pmcs_$i -- $CMD > $O/pass_$i.log 2>&1
  echo "pass $i ($SET): rc $? in $((SECONDS - T0)) s"
  DB=$(find /tmp/pmcs_$i -name "*results.db" | paste -sd, -)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py --counters "$DB" $O/pass_$i.csv "PLSVO_ALIGN_THREADS=64 staged order, bench.py --batch 8192 --steps 3 --warmup 1 (MI355X): $SET"
  [ -f $O/pass_$i.csv ] && python - <<PY
import csv, collections
rows = [r for r in csv.reader(l for l in open("$O/pass_$i.csv") if not l.startswith("#")) if len(r) == 6 and r[5] not in ("value",)]
acc = collections.defaultdict(list)
for r in rows:
    acc[r[1]].append(float(r[5]))
for k, v in acc.items():
    v = v[1:] if len(v) > 1 else v
    print("   ", k, "mean per launch", sum(v) / len(v), "launches", len(v))
PY
done
