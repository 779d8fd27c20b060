# Collect the judged artefacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh TAG      -> gpurun_out/TAG_*  (copy the ones to keep into profiles/)
cd $GRAFT_REPO_ROOT
TAG=${1:-rXX}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
rm -f $O/parity.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/${TAG}_pytest_gpu.txt
cp $O/parity.log $O/${TAG}_parity.log 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stream > $O/${TAG}_bench_under_rocprof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p2 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p3 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/p1 -name "*.db" | head -1) $O/${TAG}_kernel_stats.txt
python tools/filter_traffic.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) $O/${TAG}_filter_traffic_pmc.json > /dev/null
cp $O/${TAG}_filter_traffic_pmc.json profiles/${TAG}_filter_traffic_pmc.json      # bench.py reads the newest profiles/r*_filter_traffic_pmc.json
# where do the __amd_rocclr_copyBuffer dispatches come from?  the same trace with the library's hipEvent stage timers off
rm -rf /tmp/p4; cd /tmp; TVC_BENCH_NOTIMERS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stream > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/p4 -name "*.db" | head -1) $O/${TAG}_kernel_stats_notimers.txt
python tools/copybuffer_origin.py $(find /tmp/p4 -name "*.db" | head -1) > $O/${TAG}_copybuffer_origin.txt 2>&1
timeout 900 python bench.py 2> $O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench.json
timeout 300 python bench_stream.py 2>/dev/null | tail -1 > $O/${TAG}_stream_32streams.json      # configs[2] on its own (200 blocks)
tail -2 $O/${TAG}_pytest_gpu.txt; cut -c1-400 $O/${TAG}_bench.json; grep -c copyBuffer $O/${TAG}_kernel_stats.txt $O/${TAG}_kernel_stats_notimers.txt; grep copyBuffer $O/${TAG}_kernel_stats.txt $O/${TAG}_kernel_stats_notimers.txt
