#!/bin/bash
# usage: tools/pmc_hbm.sh OUTDIR -- python tools/prof_train.py 2      (run from the repo root on a GPU box)
# HBM bytes per kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (never combined with traces),
# plus two SQ passes for context
out=$1; shift; shift
root=$PWD; mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $root/$out/p$i -o p --output-format csv -- "${@/#tools/$root/tools}" > $root/$out/p$i.log 2>&1 || exit 1
done
