#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -x -m gpu 2>&1 | tail -15
