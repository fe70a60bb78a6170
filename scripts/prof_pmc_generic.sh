#!/bin/bash
# usage: prof_pmc_generic.sh <tag> <script.py> <kernel-name-substrings, comma separated>
# SQ counter passes (rocprofv3 --pmc with --kernel-trace only, each set in its own run) over one workload script; per-kernel summary on stdout.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
TAG=$1; SCRIPT=$2; KEYS=$3
O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"
B="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
C="FETCH_SIZE"
D="WRITE_SIZE"
for part in a b c d; do
  case $part in a) S=$A;; b) S=$B;; c) S=$C;; d) S=$D;; esac
  rocprofv3 --pmc $S --kernel-trace --output-format csv -d $O/$part -o p -- python3 $R/$SCRIPT > $O/$part.log 2>&1 || { echo "pass $part failed"; tail -3 $O/$part.log; }
done
python3 - <<PY
import csv, collections, glob, re
O="$O"; keys="$KEYS".split(",")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for part in "abcd":
    for f in glob.glob(f"{O}/{part}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(t in k for t in keys): continue
            acc[(re.search(r"k_\w+(<[^>]*>)?", k) or re.search(r".*", k)).group(0)[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    m = {n: sum(v)/len(v) for n, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0; w = max(m.get("SQ_WAVES", 1), 1)
    print(f"{k:48s} waves {w:7.0f} VALU/wave {m.get('SQ_INSTS_VALU',0)/w:8.0f} VALU-active {100*m.get('SQ_ACTIVE_INST_VALU',0)/wc:5.1f}% issue-stall {100*m.get('SQ_WAIT_INST_ANY',0)/wc:5.1f}% "
          f"waitcnt {100*m.get('SQ_WAIT_ANY',0)/wc:5.1f}% | VMEM/wave {m.get('SQ_INSTS_VMEM',0)/w:6.0f} SALU/wave {m.get('SQ_INSTS_SALU',0)/w:6.0f} "
          f"F64 fma/mul/add per wave {m.get('SQ_INSTS_VALU_FMA_F64',0)/w:.0f}/{m.get('SQ_INSTS_VALU_MUL_F64',0)/w:.0f}/{m.get('SQ_INSTS_VALU_ADD_F64',0)/w:.0f} "
          f"| FETCH {m.get('FETCH_SIZE',0)*2*1024/1e6:8.1f} MB (x2 corrected) WRITE {m.get('WRITE_SIZE',0)*1024/1e6:8.1f} MB")
PY
