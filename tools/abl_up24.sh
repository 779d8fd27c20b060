# A/B timing of the fused ups.4 kernels under different build flags (run on the GPU box).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in "$@"; do
  TVC_EXTRA_FLAGS="$f" python tinyvc_amd/build.py --force > /dev/null 2>&1
  r=$(timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['stage_ms_per_step']['filter.up4+out'],3), round(r['ms_per_step'],3))")
  echo "FLAGS=$f up4+out_ms,step_ms=$r" >> gpurun_out/abl.log
done
python tinyvc_amd/build.py --force > /dev/null 2>&1
cat gpurun_out/abl.log
