#!/bin/bash
# SQ / TCP counters of the V-cycle kernels (usage on the GPU box: tools/sq_pmc.sh TAG SIZE [extra bench args]) -> gpurun_out/TAG_sq_SIZE.txt
tag=$1; sz=$2; shift 2
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/${tag}_sq${i} --output-format csv -- python3 bench.py --size $sz --steps 3 --warmup 1 --no-cpu --no-frac512 "$@" > /dev/null 2> gpurun_out/${tag}_sq${i}.err || { tail -5 gpurun_out/${tag}_sq${i}.err; }
done
python3 tools/pmcany.py gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 gpurun_out/${tag}_sq3 gpurun_out/${tag}_sq4 > gpurun_out/${tag}_sq_${sz}.txt
rm -rf gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 gpurun_out/${tag}_sq3 gpurun_out/${tag}_sq4
grep -c . gpurun_out/${tag}_sq_${sz}.txt
