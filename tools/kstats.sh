# per-kernel stats of one bench run for a given library build (run on the GPU box):  bash tools/kstats.sh lib.so TAG
cd $GRAFT_REPO_ROOT
LIB=$1; TAG=$2
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$TAG
TVC_LIB_PATH=$GRAFT_REPO_ROOT/$LIB timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_$TAG -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stream > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/ks_$TAG -name "*.db" | head -1) gpurun_out/${TAG}_kernel_stats.txt
# optional: per-launch-position durations of one kernel:  bash tools/kstats.sh lib.so TAG PATTERN PERIOD
if [ -n "$3" ]; then python tools/kseq.py $(find /tmp/ks_$TAG -name "*.db" | head -1) "$3" "$4" | tee gpurun_out/${TAG}_kseq.txt; fi
