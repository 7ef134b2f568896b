#!/bin/bash
mkdir -p gpurun_out/r05
(time python -m pytest tests -m gpu -q) > gpurun_out/r05/tests1.txt 2>&1
