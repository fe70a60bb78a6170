#!/bin/bash
# SQ / TA counters of the p = 4 element step (scripts/prof_box_p4.py), counters in their own passes
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/pmc_box; rm -rf $O; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64"
Cc="TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o p -- python3 $R/scripts/prof_box_p4.py > $O/s.log 2>&1 || exit 1
rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/a -o p -- python3 $R/scripts/prof_box_p4.py > $O/a.log 2>&1 || exit 1
rocprofv3 --pmc $B --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/scripts/prof_box_p4.py > $O/b.log 2>&1 || echo "pass B failed"
rocprofv3 --pmc $Cc --kernel-trace --output-format csv -d $O/c -o p -- python3 $R/scripts/prof_box_p4.py > $O/c.log 2>&1 || echo "pass C failed"
python3 - <<PY
import collections, csv, glob
O = "$O"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for part in "abc":
    for f in glob.glob(f"{O}/{part}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_apply_wave" in k or "k_wave_perim" in k:
                acc["k_apply_wave" if "k_apply_wave" in k else "k_wave_perim"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{O}/s/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_apply_wave" in r["Name"] or "k_wave_perim" in r["Name"]:
            print("%-60s calls %4s avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
for k, cs in acc.items():
    m = {n: sum(v) / len(v) for n, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1.0; w = max(m.get("SQ_WAVES", 1), 1)
    print(k, "waves %.0f  per wave: VALU %.0f LDS %.0f VMEM %.0f SALU %.0f cycles %.0f" % (w, m.get("SQ_INSTS_VALU", 0)/w, m.get("SQ_INSTS_LDS", 0)/w, m.get("SQ_INSTS_VMEM", 0)/w, m.get("SQ_INSTS_SALU", 0)/w, wc/w))
    print("    of wave-cycles: VALU-active %.1f%% LDS-active %.1f%% VMEM-active %.1f%% issue-stall %.1f%% LDS-stall %.1f%% parked %.1f%%" % tuple(100*m.get(n, 0)/wc for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY")))
    print("    bank-conflict %.3g of %.3g LDS-active;  TA busy %.3g TD busy %.3g GUI active %.3g (TA busy per CU / GUI per XCD: %.2f)  TA stalled by TC %.3g" % (m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("TA_TA_BUSY_sum", 0), m.get("TD_TD_BUSY_sum", 0), m.get("GRBM_GUI_ACTIVE", 0), (m.get("TA_TA_BUSY_sum", 0)/256)/max(m.get("GRBM_GUI_ACTIVE", 1)/8, 1), m.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0)))
PY
