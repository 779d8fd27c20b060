# A/B timing with extra flags applied to ONE translation unit (run on the GPU box):
#   bash tools/abl_one.sh decoder.hip "stage1 stage2" "flags1" "flags2" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
src=$1; stages=$2; shift; shift
for f in "$@"; do
  touch tinyvc_amd/csrc/$src
  TVC_EXTRA_FLAGS="$f" python tinyvc_amd/build.py > /dev/null 2>&1
  r=$(TVC_BENCH_NOCHECK=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(' '.join(s+'='+str(round(r['stage_ms_per_step'][s],3)) for s in '$stages'.split()), 'step='+str(round(r['ms_per_step'],3)))")
  echo "FLAGS=$f $r" >> gpurun_out/abl.log
done
touch tinyvc_amd/csrc/$src
python tinyvc_amd/build.py > /dev/null 2>&1
cat gpurun_out/abl.log
