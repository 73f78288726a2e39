#!/bin/bash
for rep in 1 2; do
for v in 0 3 4 6 8 12 24; do
  PF_ZPASS_PERSIST=$v timeout 300 python scratch/zmicro.py
done
done
