#!/bin/bash
# ablation sweep of the fused forward (VLPET_DBG bits: 1 no weight stream, 2 no row loads, 4 no MFMA, 8 no stores)
for d in 0 1 2 4 8 3 7 15 11 14; do
  echo "DBG=$d: $(VLPET_DBG=$d python tools/kbench.py ${1:-28000} bf16 2>/dev/null | grep 'K1 fwd')"
done
