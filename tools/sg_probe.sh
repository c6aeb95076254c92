#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
{
for shape in 1,128,65536,16 1,256,65536,16 1,384,65536,16 2,768,32768,16 4,512,2304,16 2,512,2304,16 8,384,3152,16 4,768,8192,16 8,768,3136,16; do
  echo "== $shape default"; KB_SHAPE=$shape python tools/kbench.py fwd 2>&1 | grep scan_
  echo "== $shape sg forced"; VMS_HIP_LIB=tools/build/libvms_sgwide.so KB_SHAPE=$shape python tools/kbench.py fwd 2>&1 | grep scan_
done
} > $O/sg_probe.txt 2>&1
cat $O/sg_probe.txt
