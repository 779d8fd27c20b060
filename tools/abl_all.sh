# A/B of build flags applied to the whole library (full rebuild per variant):  bash tools/abl_all.sh "stage1 stage2" "flags1" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
stages=$1; shift
for f in "$@"; do
  TVC_EXTRA_FLAGS="$f" python tinyvc_amd/build.py --force > gpurun_out/abl_build.log 2>&1 || { echo "FLAGS=$f BUILD FAILED" >> gpurun_out/abl.log; continue; }
  r=$(timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(' '.join(s+'='+str(round(r['stage_ms_per_step'][s],3)) for s in '$stages'.split()), 'step='+str(round(r['ms_per_step'],3)))")
  echo "FLAGS=$f $r" >> gpurun_out/abl.log
done
python tinyvc_amd/build.py --force > /dev/null 2>&1
cat gpurun_out/abl.log
