#!/bin/bash
# tools/seg_sweep.sh <out> : forward / backward scan at the long-video and stack shapes with forced range counts
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
{
for s in '' 1 6 8 12 16; do
  echo "== (1,768,65536) VMS_FWD_SEGMENTS=$s"; KB_SHAPE=1,768,65536,16 VMS_DEBUG=fwd_segments=$s python tools/kbench.py fwd 2>&1 | grep scan_
done
for s in '' 8 13 16; do
  echo "== (1,768,65536) VMS_BWD_SEGMENTS=$s"; KB_SHAPE=1,768,65536,16 VMS_DEBUG=bwd_segments=$s python tools/kbench.py bwd 2>&1 | grep scan_
done
for s in '' 2 3; do
  echo "== (8,768,3136) VMS_FWD_SEGMENTS=$s"; KB_SHAPE=8,768,3136,16 VMS_DEBUG=fwd_segments=$s python tools/kbench.py fwd 2>&1 | grep scan_
done
for s in '' 2; do
  echo "== (8,768,3136) VMS_BWD_SEGMENTS=$s"; KB_SHAPE=8,768,3136,16 VMS_DEBUG=bwd_segments=$s python tools/kbench.py bwd 2>&1 | grep scan_
done
} > $O/seg_sweep.txt 2>&1
cat $O/seg_sweep.txt
