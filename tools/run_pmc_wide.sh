#!/bin/bash
# usage: tools/run_pmc_wide.sh <outdir-under-gpurun_out>   -- PMC passes for tools/pmc_conv_wide.py
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/pmc_conv_wide.py > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq3 SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_LEVEL_WAVES
