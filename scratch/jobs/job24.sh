cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j24
python -m pytest tests/test_gpu_train_graph.py tests/test_gpu_tiles.py tests/test_gpu_fov.py -x -q 2>&1 | tail -4
python tools/fov_stream.py --n-tx 3000000 --n-bd 30000 --graphed --graphed-train --train-epochs 2 --out gpurun_out/j24/fov_small.json > gpurun_out/j24/fov.log 2>&1; echo "fov_stream rc $?"; tail -2 gpurun_out/j24/fov.log | cut -c1-300
