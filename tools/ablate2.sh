#!/bin/bash
# K1 gate forward ablations (VLPET_DBG bits: 1 no global loads, 2 no stage ds_writes, 4 no down MFMAs, 8 no stores, 32 no up MFMAs, 128 no sigmoid)
for M in 2048 28000; do
for D in 0 1 2 3 4 8 32 36 128 47 175; do
  printf "M=%6d DBG=%3d " $M $D; VLPET_DBG=$D timeout 100 python tools/kbench.py $M bf16 2>&1 | grep -i "K1 fwd" | awk '{print $4, $5}'
done; done
