#!/bin/bash
# interleaved A/B of the libraries in ab/ (sweep2: chunk 128, 1 and 3 batches in flight)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in "$@"; do echo "== lib$v"; DATOK_GPU_LIB=$PWD/ab/lib$v.so timeout -k 10 120 python scripts/sweep2.py 128 1,3 2>&1 | tail -1; done
done
