#!/bin/bash
# usage: tools/run_pmc.sh <outdir-under-gpurun_out> [passes...]  -- separate --pmc passes (no other trace domains), csv output
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/pmc_conv.py > $OUT/$name.log 2>&1; }
want() { [ $# -eq 0 ] && return 0; for a in "$@"; do [ "$a" == "$CUR" ] && return 0; done; return 1; }
PASSES="$@"
for CUR in sq1 sq2 tcc1 fetch write ta tcp; do
  if [ -z "$PASSES" ] || echo " $PASSES " | grep -q " $CUR "; then
    case $CUR in
      sq1) run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS ;;
      sq2) run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT ;;
      tcc1) run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum ;;
      fetch) run fetch FETCH_SIZE ;;
      write) run write WRITE_SIZE ;;
      ta) run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum ;;
      tcp) run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum ;;
    esac
  fi
done
find $OUT -name "*.csv" | wc -l
