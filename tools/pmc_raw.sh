#!/bin/bash
# raw SQ counters of one kernel, several passes.  usage: bash tools/pmc_raw.sh <kernel-substring> -- <program args...>
K=$1; shift; shift
O=$(pwd)/gpurun_out/pmc_raw; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
P1="GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P4="SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1)); rocprofv3 --pmc $P --output-format csv -d "$O/p$i" -- "$@" > /dev/null 2> "$O/err$i.txt"; done
python3 tools/pmc_kernel.py "$O" "$K"
