# shader clock and socket power while the bench runs (GPU box)
cd $GRAFT_REPO_ROOT
(TVC_BENCH_NOCHECK=1 timeout 300 python bench.py --steps 4000 --warmup 5 --no-cpu-baseline > /tmp/b.log 2>&1 &)
sleep 30
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo; sleep 0.5; done
wait
tail -1 /tmp/b.log | cut -c1-160
