# HBM-side traffic and wait breakdown of the kNN kernel (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k1 /tmp/k2
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/k1 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d /tmp/k2 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py $(find /tmp/k1 -name "*.db" | head -1) | grep -A1 "knn_topk_split\|FilmFused\|up24s" | head -20
python tools/rocpd_pmc.py $(find /tmp/k2 -name "*.db" | head -1) | grep -A2 "knn_topk_split\|FilmFused\|up24s" | head -30
