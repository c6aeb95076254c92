#!/bin/bash
# W = 4 vs W = 8 workgroups of the backward scan at grids that emulate a two-direction launch (tools/variant.sh wk4 -DVMS_BWD_WK=4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
{
for shape in 8,768,3136,16 16,768,3136,16 8,1024,8192,16 16,1024,8192,16 2,512,2304,16; do
  for lib in "" tools/build/libvms_wk4.so; do
    echo "== $shape lib=${lib:-default} SEG=1"; env VMS_DEBUG=bwd_segments=1 ${lib:+VMS_HIP_LIB=$lib} KB_SHAPE=$shape python tools/kbench.py bwd 2>&1 | grep scan_
  done
done
} > $O/wk_probe.txt 2>&1
cat $O/wk_probe.txt
