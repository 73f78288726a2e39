#!/bin/bash
export PF_SOLVE_BESIDE_Z=0
AB_ARGS="--slab-of 8 --n 2048 --field-bytes 4 --exact-steps 0 --table-steps 0" AB_STEPS=2 bash profiles/tools/ab.sh default nospec2048 rot0 2>&1 | grep "ms per\|zpass"
