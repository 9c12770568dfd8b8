#!/bin/bash
# usage: tools/pmc_passes.sh OUTDIR -- python tools/prof_forward.py 2 1     (run from the repo root on a GPU box)
# one rocprofv3 --pmc pass per counter group (never combined with trace options)
out=$1; shift; shift
root=$PWD; mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $root/$out/p$i -o p --output-format csv -- "${@/#tools/$root/tools}" > $root/$out/p$i.log 2>&1 || exit 1
done
