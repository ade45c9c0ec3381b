"""The kernels that are no longer the default for the benchmark shapes stay correct: the same parity tests, re-run in a subprocess
with the test hook that routes through them (UR_TEST="name[=value],...", csrc/common.h: read once per process, hence the subprocess)."""
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


@pytest.mark.parametrize("env", [{"UR_TEST": "attn_no_m16"},       # the register-broadcast (VALU) kernels through the 16x16 kernels' own gate
                                 {"UR_TEST": "attn_no_mfma"}])       # no MFMA attention at all (VALU kernels, no compact rows)
def test_attention_kernel_families(env):
    _run(env, [os.path.join(HERE, "test_dropout_gpu.py"), "-k", "sasrec"], expect_min_passed=40)
    _run(env, [os.path.join(HERE, "test_gpu_parity.py"), "-k", "golden or larger_random or skip_padding"], expect_min_passed=20)


def test_gru_per_step_path_and_sorting_owner_plan():
    # gru_no_seq: no persistent recurrence kernel -- H % 64 == 0 then takes the fused step kernels (one launch per step: product + gates /
    # product + carry, round 4), other widths gemm_nt + the cell kernels; gru_no_step on top: gemm_nt + cell kernels for every width
    for env in ({"UR_TEST": "gru_no_seq"}, {"UR_TEST": "gru_no_seq,gru_no_step"}):
        _run(env, [os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_edge_cases_gpu.py"),
                   os.path.join(HERE, "test_trainer_gpu.py"), "-k", "gru or GRU or g7"], expect_min_passed=8)


def test_inline_weight_gradients():
    """UR_SASREC_SIDE=0: no side stream -- the weight-gradient launches run in line on the caller's stream (what more than 6 layers do anyway)"""
    _run({"UR_SASREC_SIDE": "0"}, [os.path.join(HERE, "test_gpu_parity.py"), "-k", "golden"], expect_min_passed=20)


def test_multi_launch_id_sort():
    """plan_multi: the multi-launch radix sort (the path of batches with more than 32 768 ids) at the small test shapes"""
    _run({"UR_TEST": "plan_multi"}, [os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_sharded.py"), "-k", "rows_plan or golden or world1"],
         expect_min_passed=20)


def test_alternative_schedules_keep_reference_parity():
    """The two-pass attention backward (default for L > 64) at L <= 64, the row-chain kernels switched all off / forward only (all on is
    the default), and a spin kernel on the side stream that widens every window in which the main stream could touch what the side stream
    has not finished with (400 us: direct readers of the parameters after step() included) -- each must pass the reference goldens and
    the oracle comparisons."""
    _run({"UR_TEST": "attn_no_m16w"}, [os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_dropout_gpu.py"), "-k", "golden or larger_random or skip_padding or attn or dropout"],
         expect_min_passed=20)
    _run({"UR_TEST": "attn_no_m16t"}, [os.path.join(HERE, "test_gpu_parity.py"), "-k", "golden or larger_random or skip_padding"], expect_min_passed=20)
    _run({"UR_TEST": "side_delay_us=400"}, [os.path.join(HERE, "test_trainer_gpu.py"), os.path.join(HERE, "test_catchup_ahead_gpu.py")], expect_min_passed=10)
    for mask in ("0", "1", "57"):
        _run({"UR_TEST": "chain_mask=" + mask}, [os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_trainer_gpu.py"),
                                         "-k", "golden or larger_random or skip_padding or sasrec or SASRec"], expect_min_passed=20)
