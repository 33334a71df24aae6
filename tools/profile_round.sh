# Regenerate the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun).
#   bash tools/profile_round.sh r02
# One --kernel-trace --stats pass per workload (kernel averages), then the PMC passes, each in its own run:
# FETCH_SIZE, WRITE_SIZE (HBM traffic, MI355X_MICROARCH.md HBM section) and the SQ counters (matrix-pipe busy).
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
B="--no-cpu-baseline --no-extra"
for wl in target c2 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$wl -- python3 $R/bench.py --workload $wl $B > $O/$wl.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  for wl in target c2 c4; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 $B > $O/pmc_${c}_$wl.log 2>&1
  done
done
for wl in target c4; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 $B > $O/pmc_sq_$wl.log 2>&1
done
if [ "$2" = "all" ] || [ "$2" = "power" ]; then
  # board power / shader clock under each workload (rocm-smi once a second during a long timed region) -> gpurun_out/power/*.txt;
  # tools/power_summary.py <tag> turns them into profiles/<tag>_power.json
  bash $R/tools/micro/power_watch.sh target "--workload target --steps 2000 --warmup 5"
  bash $R/tools/micro/power_watch.sh c4 "--workload c4 --steps 300 --warmup 2"
  bash $R/tools/micro/power_watch.sh c2 "--workload c2 --steps 15000 --warmup 10"
  cd /tmp
fi
if [ "$2" = "all" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/tools/bench_train.py --no-cpu-baseline --steps 10 > $O/train.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/plan -- python3 $R/tools/bench_plan.py --generations 1 --horizon 50 > $O/plan.log 2>&1
fi
echo done
