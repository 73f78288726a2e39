#!/bin/bash
bash scratch/pmc.sh valu SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU
bash scratch/pmc.sh busy SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
bash scratch/pmc.sh stall SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS
python scratch/pmc_summ.py valu busy stall
tail -3 gpurun_out/pmc_valu.err
