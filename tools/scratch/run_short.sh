for s in 1568,768,8 1568,768,16; do
for rep in 1 2; do
  KB_SHAPE=$s python tools/kb_short.py 2>&1 | grep "scan_bwd" | sed "s/^/lds-rs   /"
  VMS_HIP_LIB=tools/build/libvms_s_swaps.so KB_SHAPE=$s python tools/kb_short.py 2>&1 | grep "scan_bwd" | sed "s|^|swaps    |"
done; done
