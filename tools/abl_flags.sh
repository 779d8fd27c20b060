# A/B timing of one stage under different build flags (run on the GPU box):  bash tools/abl_flags.sh STAGE "flags1" "flags2" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
stage=$1; shift
for f in "$@"; do
  TVC_EXTRA_FLAGS="$f" python tinyvc_amd/build.py --force > /dev/null 2>&1
  r=$(timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['stage_ms_per_step']['$stage'],3), round(r['ms_per_step'],3))")
  echo "FLAGS=$f ${stage}_ms,step_ms=$r" >> gpurun_out/abl.log
done
python tinyvc_amd/build.py --force > /dev/null 2>&1
cat gpurun_out/abl.log
