#!/bin/bash
# UNet step time (two-stream graph replay) per library build, two rounds.  usage: ab_step_libs.sh out.txt lib1.so lib2.so ...
out=$1; shift
: > $out
for r in 1 2; do
for lib in "$@"; do
  echo "=== $lib" >> $out
  SDMI_LIB_PATH=$PWD/$lib timeout -k 10 200 python scratch/ab_step.py 2>&1 | grep "unet step" >> $out || exit 1
done
done
