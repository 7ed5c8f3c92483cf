#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_fragments_gpu.py -q -m gpu -x -k "dense" 2>&1 | tail -6
