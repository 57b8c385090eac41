#!/bin/bash
# The conv-stack PMC table alone (see tools/r05_pmc_session.sh), for the kernel as it is now: FETCH_SIZE, WRITE_SIZE and
# GRBM + MFMA busy per launch size
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05t}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PASSES=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES")
NAMES=(fetch write busy)
rm -f $O/pmc_conv_by_launch_size.csv
for cfg in "256 0" "512 0" "880 0" "1024 0" "1365 880" "1365 0" "2048 0" "4096 0"; do
  set -- $cfg; B=$1; RG=$2
  for i in 0 1 2; do
    D=$O/pmc_${B}_${RG}_${NAMES[$i]}
    if [ "$RG" != "0" ]; then export CONV_RANGE=$RG; else unset CONV_RANGE; fi
    CONV_MODES=f16x3 CONV_INPUT=boards CONV_BOARDS=$B timeout 300 rocprofv3 --output-format csv --pmc ${PASSES[$i]} -d $D -o pmc -- python $R/tools/conv_pmc.py > /dev/null 2>> $O/err.txt
  done
  python $R/tools/pmc_summary.py $O/pmc_${B}_${RG}_* | sed "s/^/${B},${RG},/" >> $O/pmc_conv_by_launch_size.csv
  rm -rf $O/pmc_${B}_${RG}_*
done
grep -E "FETCH|WRITE" $O/pmc_conv_by_launch_size.csv
