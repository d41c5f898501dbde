set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_detect.py tests/test_gpu_shapes.py -m gpu -q --no-header -p no:cacheprovider -x -k "golden or mixed_batch or c2_batch or c4_batch_shape or 720p or exact_tie or queue_overflow or best_faces or requeue" > $OUT/ab_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/ab_pytest.log
rm -f $OUT/ab_deep.txt
for rep in 1 2; do
for lib in "" alt/deepstatic.so; do
    if [ -n "$lib" ]; then export HEADTRACKR_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset HEADTRACKR_HIP_LIB; fi
    echo "lib=${lib:-product} c2 depth 3: $(timeout 120 python tools/gpu_kernel_times.py c2 "" 3 2>/dev/null | tail -1)" >> $OUT/ab_deep.txt
done
done
unset HEADTRACKR_HIP_LIB
for g in 96 128 160 256; do
    echo "product deep_grid=$g c2 depth 3: $(timeout 120 python tools/gpu_kernel_times.py c2 deep_grid=$g 3 2>/dev/null | tail -1)" >> $OUT/ab_deep.txt
done
echo "product c4 depth 2: $(timeout 120 python tools/gpu_kernel_times.py c4 "" 2 2>/dev/null | tail -1)" >> $OUT/ab_deep.txt
cat $OUT/ab_deep.txt
