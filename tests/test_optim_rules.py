"""Optimizer rules and LR schedulers of the Trainer (unirec/facility/trainer.py:134-162): the oracle's restatement and the
host-side scheduler mirrors are pinned against torch itself (the third-party library the reference delegates to); the HIP
kernels are then compared with the oracle in the -m gpu tests below."""
import math
import warnings

import numpy as np
import pytest
import torch


@pytest.mark.parametrize("wd", [0.0, 1e-2])
@pytest.mark.parametrize("algo", ["adam", "adamw", "sgd", "adagrad", "rmsprop"])
def test_oracle_rules_match_torch_optim(algo, wd):
    from oracle import model_ref
    g = torch.Generator().manual_seed(3)
    P = {"a": torch.randn(7, 5, generator=g), "b": torch.randn(11, generator=g)}
    Q = {k: torch.nn.Parameter(v.clone()) for k, v in P.items()}
    cls = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "sgd": torch.optim.SGD, "adagrad": torch.optim.Adagrad,
           "rmsprop": torch.optim.RMSprop}[algo]
    opt = cls(list(Q.values()), lr=0.05, weight_decay=wd)        # exactly how trainer.py:136-148 constructs them
    state = {}
    for step in range(6):
        G = {k: torch.randn(v.shape, generator=g) * (0.0 if (step == 3 and k == "b") else 1.0) for k, v in P.items()}   # one zero gradient
        for k in Q:
            Q[k].grad = G[k].clone()
        opt.step()
        model_ref.optimizer_step_(P, G, state, algo, 0.05, wd)
        for k in P:
            np.testing.assert_allclose(P[k].numpy(), Q[k].detach().numpy(), rtol=2e-6, atol=1e-7, err_msg=f"{algo} step {step} {k}")


class _Opt:
    def __init__(self, lr):
        self.param_groups = [{"lr": lr}]


@pytest.mark.parametrize("kind", ["step", "reduce"])
def test_scheduler_mirrors_match_torch(kind):
    from unirec_amd.facility.trainer import ReduceLROnPlateauMax, StepLRByScore
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.Adam([p], lr=0.01)
    mine = _Opt(0.01)
    if kind == "step":
        ts = torch.optim.lr_scheduler.StepLR(topt, step_size=1, gamma=0.1)
        ms = StepLRByScore(mine, 0.1)
    else:
        ts = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, mode="max", factor=0.1, patience=1, threshold=0.0001, threshold_mode="rel",
                                                        cooldown=0, min_lr=0, eps=1e-08)
        ms = ReduceLROnPlateauMax(mine, 0.1)
    scores = [0.10, 0.12, 0.11, 0.119, 0.1200001, 0.13, 0.05, 0.04, 0.03, 0.02, 1.3, 2.5, 0.2]
    for s in scores:                       # the reference calls scheduler.step(valid_score) for both kinds (trainer.py:307)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ts.step(s)
        ms.step(s)
        assert math.isclose(mine.param_groups[0]["lr"], topt.param_groups[0]["lr"], rel_tol=1e-12, abs_tol=1e-18), (kind, s)


def test_early_stopping_rule_is_the_references():
    """Trainer.early_stopping: known answers worked out from unirec/facility/trainer.py:188-233 (stop when the number of
    consecutive non-improving validations EXCEEDS early_stop for 'bigger' metrics; equality for 'smaller'; <= 0 disables)."""
    from unirec_amd.facility.trainer import Trainer
    es = Trainer.early_stopping
    assert es(0.3, None, 1, 2, True) == (0.3, 0, False, True)
    assert es(0.2, 0.3, 0, 2, True) == (0.3, 1, False, False)
    assert es(0.2, 0.3, 1, 2, True) == (0.3, 2, False, False)          # == max_step: not yet
    assert es(0.2, 0.3, 2, 2, True) == (0.3, 3, True, False)           # > max_step: stop
    assert es(0.3, 0.3, 0, 2, True) == (0.3, 1, False, False)          # a tie is not an improvement
    assert es(0.4, 0.3, 2, 2, True) == (0.4, 0, False, True)
    assert es(0.5, 0.3, 1, 2, False) == (0.3, 2, True, False)          # 'smaller': stops at equality
    assert es(0.1, 0.3, 5, 0, True) == (0.3, 5, False, True)           # disabled: always an update, nothing tracked


# ----------------------------------------------------------------------------------------------------------- HIP kernels
LR = 1e-3   # Adam-type rules move an element by ~lr whatever the size of its gradient, so rounding-noise gradients (analytically
            # zero entries) turn into +-lr differences: compared at atol = lr / 10 as in tests/test_trainer_gpu.py


def _fit(model_name, algo, wd, table_mode, n_steps=8):
    from oracle import model_ref
    from test_trainer_gpu import _setup
    from unirec_amd.facility.trainer import BatchLoader, Trainer
    from unirec_amd.utils.general import get_class_instance, init_seed
    cfg, ds = _setup(model_name, "softmax")
    cfg.update(optimizer=algo, weight_decay=wd, learning_rate=LR, embedding_optimizer=table_mode)
    init_seed(cfg["seed"])
    model = get_class_instance(model_name, "unirec_amd/model")(cfg)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr = Trainer(cfg, model)
    batches = [{k: v.cpu() for k, v in b.items()} for b in BatchLoader(ds, cfg["batch_size"], device="cuda:0")][:n_steps]
    losses = [float(tr.train_step({k: v.to("cuda:0") for k, v in b.items()})) for b in batches]
    tr.optimizer.flush()
    state = {}
    ref = [model_ref.train_step(P, state, b, cfg, lr=LR, wd=wd, algo=algo) for b in batches]
    return losses, ref, model, P


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 1e-3])
@pytest.mark.parametrize("algo", ["adamw", "sgd", "adagrad", "rmsprop", "adam"])
def test_fit_with_every_optimizer_rule_follows_dense_torch_semantics(algo, wd):
    """config['optimizer'] in {adam, adamw, sgd, adagrad, rmsprop} x weight_decay: 8 steps of the HIP trainer in the default
    lazily-evaluated dense table mode == the oracle stepping a DENSE torch-semantics optimizer over every row."""
    losses, ref, model, P = _fit("SASRec", algo, wd, "lazy_dense")
    np.testing.assert_allclose(losses, ref, rtol=2e-4)
    for k, v in model.state_dict().items():
        if k.endswith("key.bias"):
            continue
        got, ref_w = v.cpu().numpy(), P[k].numpy()
        atol = 5e-4 if algo == "rmsprop" else 1e-4     # rmsprop's early steps are 10 lr * sign-ish(g): 10x the sensitivity
        bad = np.abs(got - ref_w) > atol + 1e-3 * np.abs(ref_w)
        # the normalised rules (Adam family, Adagrad, RMSprop) step by ~lr (RMSprop's first step: 10 lr) for ANY non-zero
        # gradient, so the handful of elements whose gradient is pure rounding noise may differ by a few such steps
        assert bad.mean() <= max(1e-3, 1.5 / bad.size), (algo, k, float(bad.mean()))
        assert np.abs(got - ref_w).max() <= (0.0 if algo == "sgd" else 10 * LR * 2) + 1e-4, (algo, k)


@pytest.mark.gpu
def test_unknown_and_sparse_adam_optimizer_names():
    from test_trainer_gpu import _setup
    from unirec_amd.facility.trainer import Trainer
    from unirec_amd.utils.general import get_class_instance
    cfg, _ = _setup("MF", "bpr")
    cfg.update(optimizer="lion", weight_decay=0.1)
    tr = Trainer(cfg, get_class_instance("MF", "unirec_amd/model")(cfg))
    assert tr.optimizer.algo == "adam" and tr.optimizer.wd == 0.0          # the reference's fall-back (trainer.py:149-151)
    cfg.update(optimizer="sparse_adam")
    tr = Trainer(cfg, get_class_instance("MF", "unirec_amd/model")(cfg))
    assert tr.optimizer.algo == "adam" and tr.optimizer.table_mode == "rowwise"
