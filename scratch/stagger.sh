#!/bin/bash
# y-/z-pass rate against the byte offset between the six fields they stream side by side (PF_STAGGER)
for rep in 1 2; do
for s in 0 256 4096 69632 1052928 8392704 33558528; do
  echo -n "stagger $s: "; PF_STAGGER=$s timeout 120 python scratch/zmicro.py | cut -c1-200
done; done
