#!/bin/bash
# first-contact script for a GPU box: tests + timing probe
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -30
