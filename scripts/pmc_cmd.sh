#!/bin/bash
# Counter passes of ONE command on the GPU box (through gpurun): kernel trace, FETCH_SIZE, WRITE_SIZE, two SQ passes -- each in its own
# run, as the MI355X guide prescribes -- then a per-kernel table of the library's kernels.  usage: scripts/pmc_cmd.sh <tag> <command...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- "$@" > /dev/null 2>$OUT/trace.stderr
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o k -- "$@" > /dev/null 2>$OUT/pmc_fetch.stderr
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o k -- "$@" > /dev/null 2>$OUT/pmc_write.stderr
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU \
    --output-format csv -d $OUT/pmc_sq -o k -- "$@" > /dev/null 2>$OUT/pmc_sq.stderr
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_sq2 -o k -- "$@" > /dev/null 2>$OUT/pmc_sq2.stderr
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -type f -size +16M -delete
python scripts/pmc_table.py $TAG | tee $OUT/pmc_table.txt
