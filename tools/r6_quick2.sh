#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "tokenizer or filter_edge" 2>&1 | tail -3
