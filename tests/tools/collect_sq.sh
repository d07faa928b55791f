#!/bin/bash
# SQ counter passes of bench.py's normals workload (one counter group per pass, --kernel-trace only), reduced to a
# per-kernel table by tests/tools/reduce_sq.py.  Run on the GPU box: gpurun -- bash tests/tools/collect_sq.sh [tag]
set -u
TAG=${1:-sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS=${TWX_SQ_ARGS:-"--steps 3 --warmup 1 --no-cpu-baseline --no-daily --no-configs"}
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
P3="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P -d $OUT/p$i -o p --output-format csv -- python3 bench.py $ARGS > $OUT/p$i.log 2>&1
done
python3 tests/tools/reduce_sq.py $OUT > $OUT/sq_table.txt
cat $OUT/sq_table.txt
