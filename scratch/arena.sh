#!/bin/bash
# pass rates with the fields side by side in one allocation (PF_ARENA_FIELDS) and a pad between them (PF_ARENA_PAD)
for rep in 1 2; do
  echo -n "separate allocations: "; timeout 120 python scratch/zmicro.py | cut -c1-200
  for pad in 0 256 4096 69632 1052928 3150080; do
    echo -n "arena pad $pad: "; PF_ARENA_FIELDS=30 PF_ARENA_PAD=$pad timeout 120 python scratch/zmicro.py | cut -c1-200
  done
done
