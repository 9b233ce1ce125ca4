#!/bin/bash
# round 3: (1) vector-instruction issue rates; (2) the LDS cache of hot cells at equal chunk size, three table sizes
cd $GRAFT_REPO_ROOT
ab/valu_rate
for r in 1 2; do
  echo "== no cache, chunk 128"; DATOK_NO_HOT=1 CHUNK=128 python scripts/big_stages.py 32 2>&1 | tail -1
  for v in H H1024 H256; do
    echo "== $v chunk 128"; DATOK_GPU_LIB=$PWD/ab/lib$v.so CHUNK=128 python scripts/big_stages.py 32 2>&1 | tail -2
  done
done
