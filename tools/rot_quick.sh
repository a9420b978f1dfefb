# quick loop for the ROT extractor: parity subset, phase stamps, kernel trace (usage through gpurun: bash tools/rot_quick.sh <tag>)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-rotq}; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_extract_rot_gpu.py tests/test_reference_gpu.py tests/test_reference_cfg2_gpu.py tests/test_config0_gpu.py tests/test_frontend_frame_gpu.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o extract -- python tools/rot_phases.py > $OUT/rot_phases.log 2>&1
python tools/kstats.py $OUT/extract_kernel_stats.csv | head -8; grep -v "rocprofv3\|^W2\|^E2" $OUT/rot_phases.log | cut -c1-400
timeout 200 python tools/rot_phases.py 2>/dev/null | grep "blocking call"
