#!/bin/bash
bash tools/pmc_traffic4.sh gpurun_out/r4t "k1fwd 28000" "k1fwd 15272" "k1fwd 33200" "k1fwd 18250 192" "k1bwd 28000" "k1bwd 18250 192" "k5fwd 28000" "k5bwd 28000" "k2fwd 10000" "k2fwd 28000" "k2bwd 28000" "k3fwd 10000" "k3fwd 28000" "k3bwd 28000"
