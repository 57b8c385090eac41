R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05r; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export KSTEP_STEPS=1500 KSTEP_CACHE_LOG2=26 KSTEP_DENSE=1 KSTEP_BOARDS=1
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --output-format csv --pmc $pass -d $O/p$i -o pmc -- python $R/tools/kstep_pmc.py > $O/run$i.txt 2>> $O/err.txt
done
python $R/tools/pmc_summary.py --last=100 $O/p1 $O/p2 $O/p3 $O/p4 > $O/pmc_step_kernels_node_records.csv
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
cat $O/pmc_step_kernels_node_records.csv
