# Instruction mix and pipe-busy counters of every kernel of a bench step (run on the GPU box): gpurun_out/TAG_pmc_inst.txt
cd /tmp && export TMPDIR=/tmp
TAG=${1:-rXX}
for pass in 1 2; do
  rm -rf /tmp/k5_$pass
  if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; else C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_MFMA"; fi
  TVC_BENCH_NOCHECK=1 timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/k5_$pass -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream > /tmp/k5_$pass.log 2>&1
  tail -2 /tmp/k5_$pass.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pass in 1 2; do python tools/rocpd_pmc.py $(find /tmp/k5_$pass -name "*.db" | head -1) > gpurun_out/${TAG}_pmc_inst_$pass.txt 2>&1; done
head -30 gpurun_out/${TAG}_pmc_inst_1.txt | cut -c1-300
