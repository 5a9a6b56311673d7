#!/bin/bash
# A/B of library builds on the UNet's conv / GEMM shapes (library's own tile choice), batch 1.  usage: ab_libs.sh out.txt lib1.so lib2.so ...
out=$1; shift
: > $out
for lib in "$@"; do
  echo "=== $lib" >> $out
  SDMI_LIB_PATH=$PWD/$lib SWEEP_AUTO_ONLY=1 timeout -k 10 200 python scratch/tune_sweep.py all 1 >> $out 2>&1 || exit 1
done
