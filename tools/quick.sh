# quick GPU check: parity tests, then the bench line with stage times
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
TVC_BENCH_NOCHECK=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"
