#!/bin/bash
timeout -k 10 100 python scratch/stall_probe.py > /dev/null 2>&1
for v in base nomh hmc noscores granular; do
  VAR=$v timeout -k 10 100 python scratch/stall_probe.py 2>&1 | grep -E "^step|done|calls" | cut -c1-200
done
NP=262144 timeout -k 10 100 python scratch/stall_probe.py 2>&1 | grep -E "^step|done|calls" | cut -c1-200
NP=16777216 THRESH_MS=12 timeout -k 10 100 python scratch/stall_probe.py 2>&1 | grep -E "^step|done|calls" | cut -c1-200
