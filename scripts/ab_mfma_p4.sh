#!/bin/bash
# MFMA A/B on the p = 4 element block product: kernel time and the matrix-core counters of the block pass of a
# Chebyshev sweep on the config-5 grid, register-row form against k_blocks_residual_mfma.  Counters in their own passes (--pmc with
# --kernel-trace only).  Output: gpurun_out/mfma_p4.txt (copied to profiles/r03_mfma_p4_ab.txt).
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/mfma_p4; mkdir -p $O; out=$R/gpurun_out/mfma_p4.txt; : > $out
for m in 0 1; do
  export MIMSEM_BLOCKS_MFMA=$m
  echo "== MIMSEM_BLOCKS_MFMA=$m" >> $out
  python3 $R/scripts/prof_blocks_p4.py 2>/dev/null | tail -1 >> $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$m -o p -- python3 $R/scripts/prof_blocks_p4.py > $O/t$m.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/c$m -o p -- python3 $R/scripts/prof_blocks_p4.py > $O/c$m.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/d$m -o p -- python3 $R/scripts/prof_blocks_p4.py > $O/d$m.log 2>&1 || exit 1
  python3 - $O $m >> $out <<'PY'
import collections, csv, glob, sys
O, m = sys.argv[1], sys.argv[2]
for f in glob.glob(f"{O}/t{m}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_blocks_residual", "k_elem_apply", "k_gather_epilogue")):
            print("   %-70s calls %4s avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for part in ("c", "d"):
    for f in glob.glob(f"{O}/{part}{m}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_blocks_residual" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    v = {n: sum(x) / len(x) for n, x in cs.items()}
    print("   counters per launch of %s:" % k)
    print("      " + "  ".join("%s %.4g" % (n, v[n]) for n in sorted(v)))
    busy = v.get("SQ_BUSY_CYCLES", 0.0) or 1.0
    print("      MFMA busy / SQ busy cycles: %.3f   FP64 MFMA MOPS %.4g   FP64 VALU FMA insts %.4g" % (v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / busy, v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0), v.get("SQ_INSTS_VALU_FMA_F64", 0.0)))
PY
done
cat $out
