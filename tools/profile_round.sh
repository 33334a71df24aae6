# Regenerate the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun).
#   bash tools/profile_round.sh r01
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 $R/bench.py --no-cpu-baseline > $O/c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/target -- python3 $R/bench.py --workload target --steps 20 --warmup 3 --no-cpu-baseline > $O/target.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_c2 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/pmc_${c}_c2.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_target -- python3 $R/bench.py --workload target --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_${c}_target.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq_target -- python3 $R/bench.py --workload target --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_sq_target.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/tools/bench_train.py --no-cpu-baseline --steps 10 > $O/train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/plan -- python3 $R/tools/bench_plan.py --generations 1 --horizon 50 > $O/plan.log 2>&1
echo done
