#!/bin/bash
# Round-3 evidence in one call on one box: the default bench line, the same command under rocprofv3 --kernel-trace --stats, SEPARATE kernel
# summaries of the hot (103 680 units) and the cold (829 440 units) launch, the two PMC traffic passes, the column / Newton / HorizSolve /
# SW kernel summaries and the SQ counters of the column solves.  Outputs under gpurun_out/final3/ (copied into profiles/r03_* afterwards).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final3; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "[1] bench default"; python bench.py --steps 300 --warmup 30 > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
echo "[2] bench extras"; python bench.py --no-cpu --no-pmc --no-sweep --horiz --pcie > $O/bench_extras.json 2> $O/bench_extras.err || { tail -5 $O/bench_extras.err; exit 1; }
cd /tmp
echo "[3] rocprofv3 stats of the bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r03 -- python3 $R/bench.py --no-cpu --no-pmc --no-sw --no-column --no-families --no-sweep --steps 300 --warmup 30 > $O/bench_under_rocprof.json 2> $O/rocprof.err || { tail -5 $O/rocprof.err; exit 1; }
echo "[4] hot / cold launches separately"
for w in hot cold; do
  ONLY=$w REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o r03 -- python3 $R/scripts/prof_umat.py > $O/stats_$w.log 2>&1 || { tail -5 $O/stats_$w.log; exit 1; }
done
echo "[5] PMC traffic"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o r02 -- python3 $R/scripts/pmc_traffic.py > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o r02 -- python3 $R/scripts/pmc_traffic.py > $O/pmc_write.log 2>&1 || exit 1
cd $R
python3 scripts/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
python3 scripts/pmc_summarise.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
echo "[6] column / Newton / HorizSolve / SW kernel summaries"
cd /tmp
for s in column column3 newton horiz sw; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$s -o r03 -- python3 $R/scripts/prof_$s.py > $O/k_$s.log 2>&1 || { tail -3 $O/k_$s.log; }
done
echo "[7] SQ counters of the column solves"
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"
B="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU SQ_INSTS_VMEM"
for s in column column3; do
  rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/c_${s}_a -o p -- python3 $R/scripts/prof_$s.py > $O/c_${s}_a.log 2>&1 || exit 1
  rocprofv3 --pmc $B --kernel-trace --output-format csv -d $O/c_${s}_b -o p -- python3 $R/scripts/prof_$s.py > $O/c_${s}_b.log 2>&1 || exit 1
done
python3 - > $O/column_pmc.txt <<PY
import csv, collections, glob
O = "$O"
for s, calls in (("column", 8), ("column3", 7)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for part in ("a", "b"):
        for f in glob.glob(f"{O}/c_{s}_{part}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "at::native" in k or "rocclr" in k: continue
                acc[k[:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== scripts/prof_{s}.py: per-launch averages (percentages of SQ_WAVE_CYCLES; issue-stall = SQ_WAIT_INST_ANY, parked = SQ_WAIT_ANY) ==")
    tot = 0.0
    for k, cs in acc.items():
        m = {n: sum(v)/len(v) for n, v in cs.items()}
        nl = max(len(v) for v in cs.values())
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        fl = (2*m.get("SQ_INSTS_VALU_FMA_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_ADD_F64", 0))*64
        tot += fl*nl/calls
        print(f"{k:72s} launches {nl:4d} waves {m.get('SQ_WAVES',0):8.0f}  VALU insts/wave {m.get('SQ_INSTS_VALU',0)/max(m.get('SQ_WAVES',1),1):8.0f}  "
              f"VALU-active {100*m.get('SQ_ACTIVE_INST_VALU',0)/wc:5.1f}%  issue-stall {100*m.get('SQ_WAIT_INST_ANY',0)/wc:5.1f}%  parked {100*m.get('SQ_WAIT_ANY',0)/wc:5.1f}%  "
              f"FP64 FMA/MUL/ADD wave-insts {m.get('SQ_INSTS_VALU_FMA_F64',0):.3g}/{m.get('SQ_INSTS_VALU_MUL_F64',0):.3g}/{m.get('SQ_INSTS_VALU_ADD_F64',0):.3g}  flop/launch {fl:.3g}")
    print(f"   executed FP64 flop per solve of all 3 456 columns x 30 levels (all launches / {calls} solves in the script): {tot:.4g}   per (column, level): {tot/103680:.4g}")
PY
cat $O/column_pmc.txt | cut -c1-260
echo final profiles done
