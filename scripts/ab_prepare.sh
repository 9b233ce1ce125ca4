#!/bin/bash
# builds ab/libA.so from git HEAD (or $1) and ab/libB.so from the working tree
set -e
cd /root/repo
rev=${1:-HEAD}
mkdir -p ab /tmp/abA
rm -rf /tmp/abA/*; git archive $rev datok_amd/csrc include | tar -x -C /tmp/abA
make -C /tmp/abA/datok_amd/csrc -s OUT=/root/repo/ab/libA.so /root/repo/ab/libA.so
make -C datok_amd/csrc -s OUT=/root/repo/ab/libB.so CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result $EXTRA" /root/repo/ab/libB.so
ls -la ab/
