export KB_SHAPE=8,1024,8192,64,16
python tools/kb_cxp.py 2>&1 | grep "b, d" | sed "s/^/default /" | cut -c1-200
for v in NOBAR NOSTORE NOCONV NOMMA NOHALO xcd; do
  VMS_HIP_LIB=tools/build/libvms_cxp_$v.so python tools/kb_cxp.py 2>&1 | grep "b, d" | sed "s/^/$v /" | sed 's/.*| fused/'"$v"' fused/' | cut -c1-120
done
python tools/kb_cxp.py 2>&1 | grep "b, d" | sed 's/.*| fused/default fused/' | cut -c1-120
