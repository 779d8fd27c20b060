# FilterNet's HBM traffic alone (two PMC passes) + the bench line: bash tools/collect_traffic.sh TAG   (run on the GPU box)
cd $GRAFT_REPO_ROOT
TAG=${1:-rXX}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p2 /tmp/p3
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p2 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p3 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/filter_traffic.py $(find /tmp/p2 -name "*.db" | head -1) $(find /tmp/p3 -name "*.db" | head -1) $O/${TAG}_filter_traffic_pmc.json > /dev/null
cp $O/${TAG}_filter_traffic_pmc.json profiles/${TAG}_filter_traffic_pmc.json
timeout 900 python bench.py 2> $O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench.json
