#!/bin/bash
# second set of rocprofv3 --pmc passes over scripts/prof_umat.py: memory pipeline (TA / TCP / TCC), instruction fetch, dispatcher.
# One block type per pass where possible; counters only with --kernel-trace.  MODES="wave twopass"
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/pmc_umat2; mkdir -p $O
declare -A P
P[d]="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
P[e]="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr"
P[f]="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
P[g]="SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
P[h]="SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN"
export REPS=5
for mode in ${MODES:-wave}; do
  if [ $mode = twopass ]; then export MIMSEM_WAVE=0; else unset MIMSEM_WAVE; fi
  for k in d e f g h; do
    rocprofv3 --pmc ${P[$k]} --kernel-trace --output-format csv -d $O/${mode}_$k -o p -- python3 $R/scripts/prof_umat.py > $O/${mode}_$k.log 2>&1 || echo "pass $k failed: $(tail -2 $O/${mode}_$k.log | cut -c1-300)"
    echo "pass $mode $k done"
  done
done
python3 - <<PY
import csv, collections, glob
O="$O"
for mode in "${MODES:-wave}".split():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{mode}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(t in k for t in ("k_apply_wave", "k_elem_apply", "k_gather")):
                acc[(k.split("(")[0][-34:], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {mode} ==")
    for k, cs in sorted(acc.items()):
        print(k)
        for n, v in sorted(cs.items()):
            print(f"    {n:40s} {sum(v)/len(v):14.4g}")
PY
