#!/bin/bash
# several counter groups over one bench configuration, one rocprofv3 pass per group, each under its own timeout
# usage: pmc_groups.sh "<kernel regex>" [bench args]
cd $GRAFT_REPO_ROOT
pat=$1; shift
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
            "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "=== group $i: $ctrs"
  timeout -k 10 150 bash scripts/pmc_bench.sh g$i "$ctrs" "$@" 2>&1 | grep -E -A6 "$pat" || echo "(no output: $?)"
done
