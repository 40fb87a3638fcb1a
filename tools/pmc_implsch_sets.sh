#!/bin/bash
# diagnostics: the two SQ counter sets of the IMPLSCH cost model (issue, wait, LDS) per sea point for one kernel generation.
# usage: GEN=4 [N=131072] [PREC=sp] bash tools/pmc_implsch_sets.sh
export N=${N:-131072} PREC=${PREC:-sp} GEN=${GEN:-0}
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" TAG=a$GEN bash tools/pmc_run.sh
PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA" TAG=b$GEN bash tools/pmc_run.sh
