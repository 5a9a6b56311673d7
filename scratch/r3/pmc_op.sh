#!/bin/bash
# SQ + TCC counter passes over scratch/r3/op_only.py.  usage: pmc_op.sh out.txt filter spec [spec ...]
out=$1; flt=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmcA /tmp/pmcB /tmp/pmcC /tmp/pmcD /tmp/pmcE
run() { d=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $d -- python3 $R/scratch/r3/op_only.py "${SPECS[@]}" > /dev/null 2>&1 || exit 1; }
SPECS=("$@")
run /tmp/pmcA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run /tmp/pmcB SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS
run /tmp/pmcC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32
run /tmp/pmcD FETCH_SIZE
run /tmp/pmcE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $R
echo "# $*" > $out
PMC_FILTER=$flt python3 scratch/r3/pmc_sum.py $(find /tmp/pmcA /tmp/pmcB /tmp/pmcC /tmp/pmcD /tmp/pmcE -name "*counter_collection.csv") >> $out
