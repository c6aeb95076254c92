export KB_SHAPE=8,1024,8192,64,16
for rep in 1 2; do
python tools/kb_cxp.py 2>&1 | grep "b, d" | sed 's/.*| fused/default fused/' | cut -c1-120
for v in aux1 aux2 aux3; do
  VMS_HIP_LIB=tools/build/libvms_cxp_$v.so python tools/kb_cxp.py 2>&1 | grep "b, d" | sed 's/.*| fused/'"$v"' fused/' | cut -c1-120
done; done
