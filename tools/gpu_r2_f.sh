#!/bin/bash
# gpu_r2_e.sh + the resample phase timeline from alt/rsph.so
bash tools/gpu_r2_e.sh
LIB=headtrackr_amd/libheadtrackr_hip.so
if [ -f alt/rsph.so ]; then
  cp $LIB /tmp/base2.so; cp alt/rsph.so $LIB
  HT_RS_SUBSTAMPS=1 timeout 300 python tools/gpu_rs_phases.py c2 2>&1 | tee gpurun_out/rs_phases_c2.txt
  HT_RS_SUBSTAMPS=1 timeout 300 python tools/gpu_rs_phases.py c4 2>&1 | tee gpurun_out/rs_phases_c4.txt
  cp /tmp/base2.so $LIB
fi
