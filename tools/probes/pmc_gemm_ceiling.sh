# PMC passes over the GEMM ceiling probe (set 2: engine + two probe kernels), counters in their own runs.
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_gc_pmc/g$i -- $R/tools/probes/gemm_ceiling.bin 4096 4096 4096 1 2 > $R/gpurun_out/${TAG}_gc_pmc_g$i.log 2>&1
done
cd $R
python3 tools/pmc_shapes_summary.py gpurun_out/${TAG}_gc_pmc > gpurun_out/${TAG}_pmc_gemm_ceiling.json 2> gpurun_out/${TAG}_pmc_gemm_ceiling.err
head -c 6000 gpurun_out/${TAG}_pmc_gemm_ceiling.json
