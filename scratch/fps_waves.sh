#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scratch/fps_waves.py 2>&1 | tee gpurun_out/fps_waves.txt
