set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/j7; mkdir -p $O
python tools/host_profile.py > $O/host_profile.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr_small -- python3 tools/fov_stream.py --n-tx 10000000 --n-bd 100000 --train-batches 150 --train-epochs 3 --graphed-train > $O/tr_small.log 2>&1
python tools/prof_train_steps.py $O/tr_small 100 400 $O/small_seq.txt > $O/small_breakdown.txt 2>&1
rm -rf $O/tr_small
head -3 $O/small_breakdown.txt
