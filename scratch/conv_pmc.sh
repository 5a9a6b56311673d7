#!/bin/bash
# TCC + SQ counter passes on one conv shape.  usage: conv_pmc.sh out.txt hw cin cout batch
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc1 /tmp/pmc2 /tmp/pmc3
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d /tmp/pmc1 -- python $R/scratch/conv_only.py "$@" > /dev/null 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d /tmp/pmc2 -- python $R/scratch/conv_only.py "$@" > /dev/null 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM --output-format csv -d /tmp/pmc3 -- python $R/scratch/conv_only.py "$@" > /dev/null 2>&1 || exit 1
cd $R
echo "conv $@" > $out
PMC_FILTER=conv python scratch/pmc_sum.py $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*counter_collection.csv") >> $out
