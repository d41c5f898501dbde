set -u
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/ab_nt.txt
for rep in 1 2; do
for lib in "" alt/graynt.so; do
    if [ -n "$lib" ]; then export HEADTRACKR_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset HEADTRACKR_HIP_LIB; fi
    echo "lib=${lib:-product} c2 depth 3: $(timeout 120 python tools/gpu_kernel_times.py c2 rs_gennames=1 3 2>$OUT/ab_nt.err | tail -1)" >> $OUT/ab_nt.txt
done
done
export HEADTRACKR_HIP_LIB=$GRAFT_REPO_ROOT/alt/graynt.so
echo "lib=graynt c4 depth 2: $(timeout 120 python tools/gpu_kernel_times.py c4 rs_gennames=1 2 2>>$OUT/ab_nt.err | tail -1)" >> $OUT/ab_nt.txt
unset HEADTRACKR_HIP_LIB
echo "lib=product c4 depth 2: $(timeout 120 python tools/gpu_kernel_times.py c4 rs_gennames=1 2 2>>$OUT/ab_nt.err | tail -1)" >> $OUT/ab_nt.txt
cat $OUT/ab_nt.txt; tail -3 $OUT/ab_nt.err
