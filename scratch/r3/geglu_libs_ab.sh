#!/bin/bash
# GEGLU GEMMs per library build (SDMI_LIB_PATH), then the captured step.  usage: geglu_libs_ab.sh out.txt lib1.so lib2.so
out=$1; shift
: > $out
for r in 1 2; do
for lib in "$@"; do
  echo "=== $lib" >> $out
  SDMI_LIB_PATH=$PWD/$lib timeout -k 10 200 python scratch/r3/gemm_ab.py gemm_buf_off 2>&1 | grep "geglu" >> $out || exit 1
done
done
bash scratch/ab_step_libs.sh $out.step "$@" && cat $out.step >> $out
