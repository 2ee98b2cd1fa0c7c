#!/bin/bash
mkdir -p gpurun_out/r4p
O=gpurun_out/r4p
timeout 600 python tools/graph_probe.py --emulate-ranks 8 2>&1 | grep -v Warning | tail -30 | tee $O/graph_probe_r8.txt
timeout 600 python tools/graph_probe.py --emulate-ranks 1 2>&1 | grep -v Warning | tail -30 | tee $O/graph_probe_r1.txt
timeout 600 python tools/graph_probe.py --emulate-ranks 8 --model lora 2>&1 | grep -v Warning | tail -30 | tee $O/graph_probe_lora_r8.txt
