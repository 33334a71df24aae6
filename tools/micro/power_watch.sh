# Development: board power / clocks (rocm-smi, read-only) while a long timed region runs -- is the workload power-limited?
#   bash tools/micro/power_watch.sh "<bench.py arguments>"      e.g. "--workload c4 --steps 300 --warmup 2"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/power
python bench.py --no-extra --no-cpu-baseline $1 > gpurun_out/power/bench.txt 2>&1 &
BP=$!
for i in $(seq 1 90); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed -e 's/.*sclk clock level: .: (\([0-9]*\)Mhz)/sclk \1 MHz/' -e 's/.*Package Power (W): /power W /' | tr '\n' ' '; echo
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
