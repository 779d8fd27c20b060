# timeline dumps (S_DBG) for several flag sets:  bash tools/abl_dbg.sh "flags1" "flags2" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in "$@"; do
  touch tinyvc_amd/csrc/decoder.hip
  TVC_EXTRA_FLAGS="-DS_DBG=1 $f" python tinyvc_amd/build.py > /dev/null 2>&1
  echo "FLAGS=$f" >> gpurun_out/dbg.log
  TVC_BENCH_NOCHECK=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep sdbg | cut -c1-900 >> gpurun_out/dbg.log
done
touch tinyvc_amd/csrc/decoder.hip
python tinyvc_amd/build.py > /dev/null 2>&1
cat gpurun_out/dbg.log
