# usage (on the GPU box): bash tools/pmc_passes.sh <tag>   - four counter passes over tools/pmc_workload.py (kernel trace only beside --pmc)
TAG=${1:-pmc}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/tools/pmc_workload.py 512 > $O/p$i.log 2>&1
done
cd $R
python tools/pmc_table.py $O/p1 $O/p2 $O/p3 $O/p4 > $O/pmc_table.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
head -60 $O/pmc_table.txt
