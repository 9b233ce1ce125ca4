#!/bin/bash
# L1 (TCP) and L2 counters of the walk, one pass per group, each under its own timeout.  (Round 1 noted that "some counter groups hang rocprofv3 here" without recording which; in round 2 every group listed here completed on every run: profiles/r02_pmc_tcp.txt)
cd $GRAFT_REPO_ROOT
for g in "tcp1:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "tcp2:TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "l2:TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum"; do
  tag=${g%%:*}; ctrs=${g#*:}
  echo "=== $tag: $ctrs"
  timeout -k 10 120 bash scripts/pmc_bench.sh $tag "$ctrs" --streams 1 2>&1 | grep -A6 "k_spec_both<MatrixLean\|k_compact\|k_symbolize" || echo "(no output: $?)"
done
