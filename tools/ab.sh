#!/bin/bash
# tools/ab.sh <what: fwd|bwd> <tag>... : kernel time of each A/B library (tools/build/libvms_<tag>.so), warm clocks
what=$1; shift
for t in "$@"; do
  printf "%-14s " $t
  VMS_HIP_LIB=tools/build/libvms_$t.so python tools/kbench.py $what 2>&1 | grep scan_
done
