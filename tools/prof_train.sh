# kernel averages of the training step (rocprofv3 --kernel-trace --stats): bash tools/prof_train.sh <tag>
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_train_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/bench_train.py --no-cpu-baseline --steps 10 > $O/train.log 2>&1
find $O -name "*kernel_stats.csv" | xargs head -14 | cut -c1-160
