"""The kernels that are no longer the default for the benchmark shapes stay correct: the same parity tests, re-run in a subprocess
with the tuning switch that routes through them (the switches are read once per process, hence the subprocess)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(env, args, expect_min_passed=1):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x"] + args, env=e, capture_output=True, text=True, timeout=1500,
                       cwd=os.path.dirname(HERE))
    tail = r.stdout[-1500:] + r.stderr[-500:]
    assert r.returncode == 0, tail
    passed = [int(w) for line in r.stdout.splitlines() if " passed" in line for w in line.split() if w.isdigit()]
    assert passed and passed[0] >= expect_min_passed, tail


@pytest.mark.parametrize("env", [{"UR_ATTN_NO_M16": "1"},       # L > 64: register-broadcast kernels; L <= 64: 32x32 MFMA forward AND backward
                                 {"UR_ATTN_BWD32": "1"},        # 32x32 single-block backward for L <= 64
                                 {"UR_ATTN_NO_MFMA": "1"},      # no MFMA attention at all (VALU kernels, no compact rows)
                                 {"UR_ATTN_FWD32": "1"}])       # 32x32 single-block forward for L <= 64
def test_attention_kernel_families(env):
    _run(env, [os.path.join(HERE, "test_dropout_gpu.py"), "-k", "sasrec"], expect_min_passed=40)
    _run(env, [os.path.join(HERE, "test_gpu_parity.py"), "-k", "golden or larger_random or skip_padding"], expect_min_passed=20)


def test_gru_per_step_path_and_sorting_owner_plan():
    _run({"UR_GRU_NO_SEQ": "1"}, [os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_edge_cases_gpu.py"),
                                  os.path.join(HERE, "test_trainer_gpu.py"), "-k", "gru or GRU or g7"], expect_min_passed=8)


def test_chunked_topk_and_32_row_gemm_tiles():
    _run({"UR_TOPK_NO_PRUNE": "1"}, [os.path.join(HERE, "test_full_rank.py"), "-k", "topk and not overflow and not 3200003 and not 2200000"],
         expect_min_passed=5)
    _run({"UR_GEMM_C64": "0", "UR_SASREC_SIDE": "0"}, [os.path.join(HERE, "test_gpu_parity.py"), "-k", "golden"], expect_min_passed=20)


def test_round2_schedules_keep_reference_parity():
    """Round-2 alternatives of the default schedule: the one-workgroup id sort (the chunk-sort path is the default), the stand-alone
    LayerNorm-backward launches (the fused GEMM epilogue is the default), and the row-chain kernels switched all off / all on (forward only is the default) -- each must pass the
    reference goldens and the oracle comparisons."""
    _run({"UR_PLAN_ONEWG": "1"}, [os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_sharded.py"), "-k", "rows_plan or golden or world1"],
         expect_min_passed=20)
    _run({"UR_SASREC_NO_LNFUSE": "1"}, [os.path.join(HERE, "test_gpu_parity.py"), "-k", "golden or larger_random or skip_padding"], expect_min_passed=20)
    # no chain kernels at all / every chain kernel (default: forward chain + both chains of the last-row layer) / round 2a's default /
    # the gemm_tn variants (LDS-staged is the default; the no-LDS kernel with an 8- or 16-deep register ring)
    _run({"UR_TN_DIRECT": "8"}, [os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_gemm_gpu.py"), "-k", "golden or larger_random or tn"], expect_min_passed=20)
    _run({"UR_TN_DIRECT": "16"}, [os.path.join(HERE, "test_gemm_gpu.py"), "-k", "tn"], expect_min_passed=10)
    # the two-pass, workgroup-per-head attention backward (the wave-per-head single-pass kernel is the default for L <= 64)
    _run({"UR_ATTN_NO_M16W": "1"}, [os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_dropout_gpu.py"), "-k", "golden or larger_random or skip_padding or attn or dropout"],
         expect_min_passed=20)
    # the last-row layer as ONE workgroup per row block (default: the inner dimension split over workgroups)
    _run({"UR_SASREC_NO_QFUSE": "1"}, [os.path.join(HERE, "test_gpu_parity.py"), "-k", "golden or larger_random or skip_padding"], expect_min_passed=20)   # gather + query GEMM launches in front of the one-query attention
    _run({"UR_SASREC_NO_SPLIT": "1"}, [os.path.join(HERE, "test_gpu_parity.py"), "-k", "golden or larger_random or skip_padding"], expect_min_passed=20)
    # the dense half of the optimizer step on the main stream (round 2a) / on the side stream but joined by step() itself; a late join in
    # front of the next forward pass's first launch
    # ({"UR_SASREC_STOP_EVENTS": "0"}: every fork of the backward pass by hipEventRecord instead of an event carried by the producing launch)
    # ({"UR_SASREC_SAVE_U": "1"}: the forward chain keeps act(h1) for the FFN-2 weight gradient instead of recomputing it there;
    #  {"UR_SASREC_EARLY_REDUCE": "0"}: every deferred reduction at the end of the pass)
    for env in ({"UR_DENSE_ADAM_SIDE": "0"}, {"UR_DENSE_ADAM_SIDE": "join"}, {"UR_SIDE_JOIN_TOP": "1"}, {"UR_SASREC_STOP_EVENTS": "0"},
                {"UR_SASREC_SAVE_U": "1"}, {"UR_SASREC_EARLY_REDUCE": "0"},
                # a spin kernel in front of the dense half on the side stream: every window in which the main stream could touch what
                # the side stream has not finished with is 400 us wide (direct readers of the parameters after step() included)
                # (not together with UR_DENSE_ADAM_SIDE=late: that override leaves the join to the next forward pass even for the last
                # step, and these tests read the parameters right after it -- the case step(late_join=False) exists for)
                {"UR_SIDE_TEST_DELAY_US": "400"}):
        _run(env, [os.path.join(HERE, "test_trainer_gpu.py"), os.path.join(HERE, "test_catchup_ahead_gpu.py")], expect_min_passed=10)
    for mask in ("0", "63", "1"):
        _run({"UR_SASREC_CHAIN": mask}, [os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_trainer_gpu.py"),
                                         "-k", "golden or larger_random or skip_padding or sasrec or SASRec"], expect_min_passed=20)
