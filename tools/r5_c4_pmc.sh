#!/bin/bash
# config-4 step kernel: where the table rows come from (L2 / MALL counters) and how busy the vector pipe is; one counter pass each
cd ${GRAFT_REPO_ROOT:-$(pwd)}
A="--ndim 1000 --nwalkers 512 --steps 10 --warmup 12"
bash tools/pmc_one.sh r5c4_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_READ_sum" $A
bash tools/pmc_one.sh r5c4_tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" $A
bash tools/pmc_one.sh r5c4_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" $A
bash tools/pmc_one.sh r5c4_sq2 "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM" $A
bash tools/pmc_one.sh r5c4_ta "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" $A
tail -3 gpurun_out/r5c4_*.log | cut -c1-300
