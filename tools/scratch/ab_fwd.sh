for rep in 1 2 3; do
  python tools/kbench.py fwd 2>&1 | grep scan_fwd | sed "s/^/new  /"
  VMS_HIP_LIB=tools/build/libvms_f_nobuf.so python tools/kbench.py fwd 2>&1 | grep scan_fwd | sed "s/^/old  /"
done
KB_SHAPE=8,768,3136,16 python tools/kbench.py fwd 2>&1 | grep scan_fwd | sed "s/^/new 3136 /"
KB_SHAPE=8,768,3136,16 VMS_HIP_LIB=tools/build/libvms_f_nobuf.so python tools/kbench.py fwd 2>&1 | grep scan_fwd | sed "s/^/old 3136 /"
