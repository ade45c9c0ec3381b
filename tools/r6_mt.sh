#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_data_path.py -q -x -m gpu -k "mt_device" -s 2>&1 | tail -25
