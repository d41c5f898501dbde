cd $GRAFT_REPO_ROOT
for v in product h512 h1024 h1024b h256u8; do
  L=""; [ $v != product ] && L=$GRAFT_REPO_ROOT/alt/$v.so
  for f in 8 1; do HEADTRACKR_HIP_LIB=$L timeout 200 python tools/gpu_cs_step.py $f 2>/dev/null | tail -1; done
done
