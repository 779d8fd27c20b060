# A/B of knn.hip build flags (run on the GPU box): bash tools/abl_knn.sh "flags1" "flags2" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in "$@"; do
  touch tinyvc_amd/csrc/knn.hip
  TVC_EXTRA_FLAGS="$f" python tinyvc_amd/build.py > /dev/null 2>&1
  r=$(timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('knn='+str(round(r['stage_ms_per_step']['knn'],3)), 'step='+str(round(r['ms_per_step'],3)))")
  echo "FLAGS=$f $r" >> gpurun_out/abl.log
done
touch tinyvc_amd/csrc/knn.hip
python tinyvc_amd/build.py > /dev/null 2>&1
cat gpurun_out/abl.log
