#!/usr/bin/env python3
"""Runs a few GEMM launches for PMC collection (rocprofv3 --pmc ... -- python tools/gemm_pmc.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import nt, tn
M = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
nt(M, 384, 128, 0, 1)
nt(M, 128, 512, 1, 2)
tn(M, 128, 512, 1)
