# MFMA-busy / LDS / wait counters of every kernel of a bench step (run on the GPU box): gpurun_out/TAG_pmc_mfma_lds.txt
cd /tmp && export TMPDIR=/tmp
TAG=${1:-rXX}
rm -rf /tmp/k3
TVC_BENCH_NOCHECK=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/k3 -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/rocpd_pmc.py $(find /tmp/k3 -name "*.db" | head -1) > gpurun_out/${TAG}_pmc_mfma_lds.txt
head -40 gpurun_out/${TAG}_pmc_mfma_lds.txt | cut -c1-260
