#!/bin/bash
# knock-out builds of the working tree into ab/: libB.so (as is), libK<n>.so with -DDTK_KO=<n> -DDTK_EXPERIMENTS
set -e
cd /root/repo
mkdir -p ab
F="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DDTK_EXPERIMENTS"
make -C datok_amd/csrc -s OUT=/root/repo/ab/libB.so CXXFLAGS="$F" /root/repo/ab/libB.so
for k in "$@"; do
  make -C datok_amd/csrc -s OUT=/root/repo/ab/libK$k.so CXXFLAGS="$F -DDTK_KO=$k" /root/repo/ab/libK$k.so
done
ls -la ab/*.so
