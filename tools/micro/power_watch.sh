# Development: board power / clocks (rocm-smi, read-only) while a long timed region runs -- is the workload power-limited?
#   bash tools/micro/power_watch.sh <name> "<bench.py arguments>"      e.g. c4 "--workload c4 --steps 300 --warmup 2"
# Samples go to gpurun_out/power/<name>.txt (one line per second: "power W <w> sclk <mhz> MHz"), bench.py's line to <name>.bench.txt;
# tools/power_summary.py turns them into profiles/<tag>_power.json (read by bench.py, digest-checked).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/power
NAME=$1
python bench.py --no-extra --no-cpu-baseline $2 > gpurun_out/power/$NAME.bench.txt 2>&1 &
BP=$!
: > gpurun_out/power/$NAME.txt
for i in $(seq 1 120); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed -e 's/.*sclk clock level: .*(\([0-9]*\)Mhz)/sclk \1 MHz/' -e 's/.*Package Power (W): /power W /' | tr '\n' ' ' >> gpurun_out/power/$NAME.txt; echo >> gpurun_out/power/$NAME.txt
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" > gpurun_out/power/cap.txt || true
