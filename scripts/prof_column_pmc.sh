#!/bin/bash
# PMC passes (rocprofv3 --pmc with --kernel-trace only; counters in their own runs) over the column Schur sweep, scripts/prof_column.py:
# the fused DPP kernels of round 2 (default) and the round-1 row-per-lane kernels (MIMSEM_SCHUR_FUSED=rows) side by side.
# Output: gpurun_out/pmc_col/{sweep,rows}_{a,b}/ and a per-kernel summary on stdout (copied to profiles/r02_column_pmc.txt).
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/pmc_col; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"
B="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS"
for mode in ${MODES:-sweep rows}; do
  export MIMSEM_SCHUR_FUSED=$mode
  rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/${mode}_a -o p -- python3 $R/scripts/prof_column.py > $O/${mode}_a.log 2>&1 || exit 1
  rocprofv3 --pmc $B --kernel-trace --output-format csv -d $O/${mode}_b -o p -- python3 $R/scripts/prof_column.py > $O/${mode}_b.log 2>&1 || exit 1
done
python3 - <<PY
import csv, collections, glob
O="$O"
for mode in "${MODES:-sweep rows}".split():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for part in ("a", "b"):
        for f in glob.glob(f"{O}/{mode}_{part}/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if not any(t in k for t in ("k_schur", "k_thomas", "k_block_thomas", "k_coef_block")): continue
                acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== MIMSEM_SCHUR_FUSED={mode}: per-launch averages over the dispatches of scripts/prof_column.py ==")
    for k, cs in acc.items():
        m = {n: sum(v)/len(v) for n, v in cs.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        line = f"{k:70s} waves {m.get('SQ_WAVES',0):9.0f}  VALU insts/wave {m.get('SQ_INSTS_VALU',0)/max(m.get('SQ_WAVES',1),1):8.0f}  " \
               f"VALU-active {100*m.get('SQ_ACTIVE_INST_VALU',0)/wc:5.1f}%  issue-stall {100*m.get('SQ_WAIT_INST_ANY',0)/wc:5.1f}%  " \
               f"parked(waitcnt) {100*m.get('SQ_WAIT_ANY',0)/wc:5.1f}% of wave-cycles;  LDS insts/wave {m.get('SQ_INSTS_LDS',0)/max(m.get('SQ_WAVES',1),1):7.0f}  " \
               f"LDS bank-conflict cycles {m.get('SQ_LDS_BANK_CONFLICT',0):.3g} of {m.get('SQ_LDS_IDX_ACTIVE',0):.3g} active;  " \
               f"FP64 FMA/MUL/ADD insts {m.get('SQ_INSTS_VALU_FMA_F64',0):.3g}/{m.get('SQ_INSTS_VALU_MUL_F64',0):.3g}/{m.get('SQ_INSTS_VALU_ADD_F64',0):.3g}"
        print(line)
PY
