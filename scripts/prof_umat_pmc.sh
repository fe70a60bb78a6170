#!/bin/bash
# rocprofv3 passes over scripts/prof_umat.py (the bench.py step alone): kernel stats, SQ counters and HBM traffic of the wave-level
# fused kernel (default) and of the two-pass form (MIMSEM_WAVE=0).  Counters in their own runs, --kernel-trace only.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/pmc_umat; mkdir -p $O
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64"
for mode in ${MODES:-wave twopass}; do
  if [ $mode = twopass ]; then export MIMSEM_WAVE=0; else unset MIMSEM_WAVE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${mode}_s -o p -- python3 $R/scripts/prof_umat.py > $O/${mode}_s.log 2>&1 || exit 1
  rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/${mode}_a -o p -- python3 $R/scripts/prof_umat.py > $O/${mode}_a.log 2>&1 || exit 1
  rocprofv3 --pmc $B --kernel-trace --output-format csv -d $O/${mode}_b -o p -- python3 $R/scripts/prof_umat.py > $O/${mode}_b.log 2>&1 || echo "pass B failed"
  # HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md; together they crash the profiler on this pool)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${mode}_f -o p -- python3 $R/scripts/prof_umat.py > $O/${mode}_f.log 2>&1 || echo "pass F failed"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${mode}_w -o p -- python3 $R/scripts/prof_umat.py > $O/${mode}_w.log 2>&1 || echo "pass W failed"
  echo "mode $mode done"
done
python3 $R/scripts/pmc_umat_summary.py $O
