for s in 1568,768,8 1568,768,16 1568,768,4; do
for lib in default tools/build/libvms_s_nolds.so default tools/build/libvms_s_nolds.so; do
  if [ $lib = default ]; then KB_SHAPE=$s python tools/kb_short.py 2>&1 | grep "scan_" | sed "s/^/new   /"; else VMS_HIP_LIB=$lib KB_SHAPE=$s python tools/kb_short.py 2>&1 | grep "scan_" | sed "s|^|old   |"; fi
done; done
