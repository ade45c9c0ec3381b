"""Trainer -- mirror of unirec/facility/trainer.py:21-458 for the accelerated path.

Same constructor / method names (``Trainer(config, model, accelerator)``, ``fit``, ``evaluate``, ``save_model``,
``load_model``, ``set_user_history``, ``reset_evaluator``) and the same checkpoint dict keys
(``config, cur_epoch, cur_step, best_valid_score, state_dict, optimizer, scheduler``: trainer.py:390-398).
The loop body (trainer.py:327-357) becomes: plan ids -> forward -> backward -> clip -> step, with the loss kept on
the device and read back once per epoch instead of two host syncs per step."""
import logging
import collections
import math
import os
import time

import numpy as np
import torch

from .distributed import ShardedSparseDenseAdam, dist_group, dist_info
from .optimizer import SparseDenseAdam


class BatchLoader:
    """Index batches over a unirec_amd dataset; rank r takes batches r, r+W, ... (Accelerate's default sharding,
    SURVEY.md Appendix B), the last partial batch is kept."""

    def __init__(self, dataset, batch_size, shuffle=False, seed=2022, rank=0, world=1, device="cuda:0"):
        self.dataset, self.batch_size, self.shuffle, self.seed = dataset, batch_size, shuffle, seed
        self.rank, self.world, self.device, self.epoch = rank, world, torch.device(device), 0
        # 1: one ``random()`` is drawn from the row builder's stream behind every batch.  The reference's BPR / CCL loss draws
        # ``random.random()`` once per training forward (its 10 % label check, unirec/model/base/reco_abc.py:238-246) from the process-global
        # stream its negative sampler and history cut use, i.e. in the middle of the row stream (__iter__ says where): Trainer.fit sets this
        # for those losses, so the sampled negatives stay the reference's over a whole run (tests/golden/g10_trainer_fit_{gru,mf_c1})
        self.loss_check_draws = 0

    def __len__(self):
        nb = (len(self.dataset) + self.batch_size - 1) // self.batch_size
        return (nb + self.world - 1) // self.world

    def __iter__(self):
        n = len(self.dataset)
        order = np.random.default_rng(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        self.epoch += 1
        nb = (n + self.batch_size - 1) // self.batch_size
        # every rank takes the same number of steps (the step is collective): a rank whose share runs out wraps around to
        # the first batches, as Accelerate's even_batches default does
        draws = bool(self.loss_check_draws) and hasattr(self.dataset, "_builder")
        held = None
        for k in range((nb + self.world - 1) // self.world):
            b = (k * self.world + self.rank) % nb
            rows = self.dataset.get_batch(order[b * self.batch_size:(b + 1) * self.batch_size])
            batch = {k: torch.from_numpy(v).to(self.device, non_blocking=True) for k, v in rows.items()}
            if not draws:
                yield batch
                continue
            # the reference trains through Accelerate's DataLoaderShard, which builds batch i + 1 BEFORE it hands out batch i (it looks one
            # batch ahead to flag the last one), so the loss's draw of step i lands between the rows of batch i + 1 and those of batch i + 2
            if held is not None:
                yield held
                self.dataset._builder().random()
            held = batch
        if held is not None:
            yield held
            self.dataset._builder().random()


class DeviceBatchLoader:
    """Device-resident input pipeline: the (user, item) interaction pairs live in HBM and every batch
    is built there by ``DeviceRowBuilder`` (negatives, history cut, left padding: two launches, no host work, no H2D
    copy).  Sharding and last-partial-batch behaviour as ``BatchLoader``.

    The builds run on the loader's OWN stream, ``lookahead`` batches ahead of the consumer: a batch is handed out behind a wait for an
    event that completed a step ago, so the half-dozen small launches of a build (index, two column copies, sampler, history cut) sit
    beside the training step's kernels instead of in front of them (bench.py `e2e`; at the headline shape the builds are ~1 % of the step
    either way -- what separates `e2e` from the headline line is the lazy-Adam replay of rows seen before and 1 % more real tokens)."""

    def __init__(self, pairs, builder, batch_size, shuffle=False, seed=2022, rank=0, world=1, with_seq=True, lookahead=2):
        dev = builder.device
        self.pairs = (pairs if torch.is_tensor(pairs) else torch.from_numpy(np.asarray(pairs).astype(np.int64))).to(dev).contiguous()
        self.builder, self.batch_size, self.shuffle, self.seed = builder, batch_size, shuffle, seed
        self.rank, self.world, self.with_seq, self.epoch = rank, world, with_seq, 0
        self.lookahead = max(1, int(lookahead))
        self.stream = torch.cuda.Stream(device=self.pairs.device) if self.pairs.is_cuda else None
        self.joined = False

    def use_stream(self, stream, joined=False):
        """Build on `stream` instead of a stream of the loader's own.  Trainer.fit hands over the optimizer's PLAN stream (the id sort of the
        next batch runs there, under the step in flight): a process has a handful of hardware queues and HIP deals streams onto them
        round-robin -- the loader's own stream landed on the queue of the encoder's side stream (weight-gradient launches, dense update),
        its half-dozen small kernels queued among them, and the step boundary waited 60-90 us for that chain (profiles/r05_e_fit_gap.txt).

        joined=True is a promise of the consumer that saves the two packets a hand-out otherwise puts into ITS queue (a barrier and an
        event, ~8 us of idle main stream per step): (a) before it reads a batch other than the first of an epoch, its stream has waited for
        an event recorded on `stream` after that batch was handed out; (b) once per step, before it enqueues anything on `stream`, it makes
        `stream` wait for its own stream.  SparseDenseAdam.prefetch_plan / plan_batch do both for a batch passed as `next_batch`."""
        if stream is not None and self.pairs.is_cuda:
            self.stream, self.joined = stream, bool(joined)

    def __len__(self):
        nb = (len(self.pairs) + self.batch_size - 1) // self.batch_size
        return (nb + self.world - 1) // self.world

    def _build(self, order, b, base):
        B = self.batch_size
        sel = self.pairs[order[b * B:(b + 1) * B]]
        return self.builder.build(sel[:, 0].contiguous(), sel[:, 1].contiguous(), with_seq=self.with_seq, step=base + b)

    def _epoch_columns(self, order):
        """the epoch's (user, item) columns in visiting order, [2, n] contiguous: every batch is then two VIEWS (no per-batch index gather
        and column copies: three launches and ~30 us of host time a step); made once per epoch, or once for good without shuffling"""
        if not self.shuffle:
            if getattr(self, "_cols", None) is None:
                self._cols = self.pairs.t().contiguous()
            return self._cols
        return self.pairs[order].t().contiguous()

    def _build_cols(self, cols, b, base):
        B = self.batch_size
        return self.builder.build(cols[0, b * B:(b + 1) * B], cols[1, b * B:(b + 1) * B], with_seq=self.with_seq, step=base + b)

    def __iter__(self):
        n, B = len(self.pairs), self.batch_size
        dev = self.pairs.device
        if self.shuffle:
            g = torch.Generator(device=dev).manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g, device=dev)
        else:
            order = torch.arange(n, device=dev)
        nb = (n + B - 1) // B
        base = self.epoch * nb
        self.epoch += 1
        ks = iter(range((nb + self.world - 1) // self.world))   # equal step counts on every rank (see BatchLoader)
        cols = self._epoch_columns(order)
        if self.stream is None:
            for k in ks:
                yield self._build_cols(cols, (k * self.world + self.rank) % nb, base)
            return
        self.stream.wait_stream(torch.cuda.current_stream(dev))   # (`order` / `cols` were made on the caller's stream)
        pending = collections.deque()
        # Batches are made on the loader's stream and read on the caller's.  Tensor.record_stream would keep the allocator from recycling
        # them early, but it costs one event packet IN THE CALLER'S QUEUE per tensor when the batch is dropped (five per step, each a
        # marker the step's next kernel queues behind: ~25 us of idle main stream per 0.55 ms step, profiles/r05_e_fit_gap.txt).  Instead:
        # at every hand-out the loader's stream waits for the caller's stream (one fence-less event), so any build issued from then on
        # runs behind every step the caller has enqueued so far; a handed-out batch stays referenced here for two more hand-outs, by which
        # time the step that read it is enqueued, and its memory (the loader stream's pool) can only be recycled by such a later build.
        # (joined: the consumer's own once-per-step wait orders this stream one step later than a wait here would: one more batch held)
        handed = collections.deque(maxlen=4 if self.joined else 3)

        def issue():
            k = next(ks, None)
            if k is None:
                return
            with torch.cuda.stream(self.stream):
                batch = self._build_cols(cols, (k * self.world + self.rank) % nb, base)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            pending.append((batch, ev))

        for _ in range(self.lookahead):
            issue()
        from .. import ops
        try:
            while pending:
                batch, ev = pending.popleft()
                cur = torch.cuda.current_stream(dev)
                if not (self.joined and handed):
                    cur.wait_event(ev)
                handed.append(batch)
                if not self.joined:
                    ops.stream_wait_stream(self.stream, cur)
                issue()
                yield batch
        finally:
            # the last batches (and an iteration abandoned half way): the caller may still be reading them when the generator goes away
            for b in list(handed) + [b for b, _ in pending]:
                for t in b.values():
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(torch.cuda.current_stream(dev))


class StepLRByScore:
    """What the reference's 'step' scheduler does: StepLR(step_size=1, gamma=factor) stepped as ``scheduler.step(valid_score)``
    (trainer.py:156,307).  StepLR.step(epoch) with an explicit argument takes the closed form
    lr = base_lr * gamma ** (epoch // step_size) -- and the argument is the validation score, so the exponent is
    floor(valid_score): the learning rate stays at base_lr for every metric below 1."""

    def __init__(self, optimizer, gamma):
        self.optimizer, self.gamma = optimizer, gamma
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = 0

    def step(self, epoch):
        self.last_epoch = math.floor(epoch)
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = base * self.gamma ** (self.last_epoch // 1)

    def state_dict(self):
        return {"gamma": self.gamma, "base_lrs": self.base_lrs, "last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.gamma, self.base_lrs, self.last_epoch = sd["gamma"], list(sd["base_lrs"]), sd["last_epoch"]


class ReduceLROnPlateauMax:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='max', factor, patience=1, threshold=1e-4 'rel', cooldown=0, min_lr=0,
    eps=1e-8) as built at trainer.py:158-159."""

    def __init__(self, optimizer, factor, patience=1, threshold=1e-4, min_lr=0.0, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.optimizer, self.factor, self.patience, self.threshold, self.min_lr, self.eps = optimizer, factor, patience, threshold, min_lr, eps
        self.best, self.num_bad_epochs, self.last_epoch = -math.inf, 0, 0

    def step(self, metrics):
        current = float(metrics)
        self.last_epoch += 1
        if current > self.best * (self.threshold + 1.0):
            self.best, self.num_bad_epochs = current, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            for g in self.optimizer.param_groups:
                old = float(g["lr"])
                new = max(old * self.factor, self.min_lr)
                if old - new > self.eps:
                    g["lr"] = new
            self.num_bad_epochs = 0

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class Trainer(object):
    def __init__(self, config, model, accelerator=None):
        self.config, self.model, self.accelerator = config, model, accelerator
        # one process per GPU (unirec/facility/trainer.py:67,261: accelerator.prepare wraps model / optimizer / loaders): with
        # world > 1 the optimizer built below trains ONE model -- tables row-sharded, dense parameters replicated and
        # all-reduced (facility/distributed.py); world == 1 is the plain single-GPU path
        self.rank, self.world = dist_info(accelerator)
        self.exp_name = config.get("exp_name", __name__)
        self.logger = logging.getLogger(self.exp_name)
        self.learning_rate = config.get("learning_rate", 1e-3)
        self.epochs = config.get("epochs", 0)
        self.early_stop = config.get("early_stop", 0)
        self.weight_decay = config.get("weight_decay", 0)
        self.key_metric = config.get("key_metric", "hit@5")
        self.checkpoint_dir = os.path.join(config.get("output_path", "./output"), config.get("checkpoint_dir", "checkpoint"))
        self.saved_model_file = os.path.join(self.checkpoint_dir, f"{self.exp_name}.pth")
        gc = config.get("grad_clip_value", None)
        self.grad_clip_value = gc if gc and gc > 0 else None
        self.optimizer = self._build_optimizer(config.get("optimizer", "adam"))
        self.scheduler = self._build_scheduler(config.get("scheduler", "off"), config.get("scheduler_factor", 0.1))
        self.best_valid_score, self.cur_step, self.start_epoch = None, 1, 0
        self.step_losses = []

    def _build_optimizer(self, opt_type):
        """Trainer._build_optimizer (unirec/facility/trainer.py:134-152): same names, same fall-back."""
        table_mode = self.config.get("embedding_optimizer", "lazy_dense")
        wd = self.weight_decay
        if opt_type == "sparse_adam":
            # torch.optim.SparseAdam refuses the dense gradients the reference's nn.Embedding produces; its meaning -- Adam that
            # only moves the rows a batch touches, no weight decay -- is this backend's row-wise table mode
            if wd > 0:
                self.logger.warning("Sparse Adam cannot argument received argument [weight_decay]")
            opt_type, table_mode, wd = "adam", "rowwise", 0.0
        elif opt_type not in ("adam", "sgd", "adagrad", "rmsprop", "adamw"):
            self.logger.warning("Received unrecognized optimizer, set default Adam optimizer")
            opt_type, wd = "adam", 0.0          # the reference's fall-back drops weight_decay too (trainer.py:151)
        if self.world > 1:
            return ShardedSparseDenseAdam(self.model, self.rank, self.world, group=dist_group(self.accelerator), lr=self.learning_rate,
                                          weight_decay=wd, grad_clip=self.grad_clip_value, table_mode=table_mode, algo=opt_type)
        return SparseDenseAdam(self.model, lr=self.learning_rate, weight_decay=wd, grad_clip=self.grad_clip_value, table_mode=table_mode,
                               algo=opt_type)

    def _build_scheduler(self, scheduler_type, factor):
        """Trainer._build_scheduler (trainer.py:154-162); stepped after each validation of epochs > 0 with the validation score
        (trainer.py:306-307)."""
        if scheduler_type == "step":
            return StepLRByScore(self.optimizer, factor)
        if scheduler_type == "reduce":
            return ReduceLROnPlateauMax(self.optimizer, factor)
        return None

    def set_user_history(self, user_history):
        self.user_history = user_history

    def reset_evaluator(self, data_format=None, eval_protocol=None):
        self.eval_protocol = eval_protocol

    # ------------------------------------------------------------------ the step (trainer.py:327-357)
    def _plan_ids(self, batch):
        return dict(item_seq=batch.get("item_seq"), item_id=batch["item_id"],
                    user_id=batch.get("user_id") if hasattr(self.model, "user_embedding") else None)

    def train_step(self, batch, next_batch=None):
        """One optimisation step.  `next_batch` (optional) lets the id sort of the following batch run on a side
        stream underneath this step's forward/backward."""
        opt, model = self.optimizer, self.model
        model.train()
        if self.world > 1:     # the collective step: plans, row exchange, forward / backward, gradient exchange, updates
            return opt.train_step(batch, next_batch)
        opt.zero_grad()
        opt.plan_batch(**self._plan_ids(batch))
        if next_batch is not None:
            opt.prefetch_plan(**self._plan_ids(next_batch))
        kw = {k: batch[k] for k in ("user_id", "item_id", "label", "item_seq", "item_seq_len") if k in batch}
        if self.config.get("fused_step", True):
            loss = model.forward_backward(**kw)       # same launches as forward + backward, no autograd graph
        else:
            loss, _, _, _ = model(**kw)
            loss.backward()
        opt.step(late_join=next_batch is not None)   # (another step follows at once: its forward pass joins the side-stream half of this update)
        return loss.detach()

    @staticmethod
    def early_stopping(value, best, cur_step, max_step=4, bigger=True):
        """Trainer.early_stopping (unirec/facility/trainer.py:188-233), value for value: -> (best, cur_step, stop_flag, update_flag).
        Note the asymmetry the reference has: 'bigger' stops when cur_step > max_step, 'smaller' when cur_step >= max_step; with
        max_step <= 0 early stopping is off and every validation counts as an update."""
        stop_flag = update_flag = False
        if max_step > 0:
            if (best is None) or (value > best if bigger else value < best):
                cur_step, best, update_flag = 0, value, True
            else:
                cur_step += 1
                stop_flag = cur_step > max_step if bigger else cur_step >= max_step
        else:
            update_flag = True
        return best, cur_step, stop_flag, update_flag

    def fit(self, train_data, valid_data=None, save_model=True, load_pretrained_model=False, model_file=None, verbose=2):
        if load_pretrained_model:
            if model_file is None:
                raise ValueError("`model_file` should be given when `load_pretrained_model` is set to True.")
            self.load_model(model_file)
        # everything alive now (torch, the model, the datasets) goes to the collector's permanent generation: a full collection of a process
        # that has imported torch is a ~35 ms host pause, and a 0.7 ms step has only a few ms of enqueued work to cover it
        import gc
        gc.collect()
        gc.freeze()
        for epoch_idx in range(self.start_epoch, self.epochs):
            if valid_data is not None:
                res = self.evaluate(valid_data, load_best_model=False)
                score = res[self.key_metric]
                self.best_valid_score, self.cur_step, stop_flag, update_flag = Trainer.early_stopping(
                    score, self.best_valid_score, self.cur_step, max_step=self.early_stop, bigger=True)
                self.logger.info("epoch %d evaluating [%s: %f]", epoch_idx, self.key_metric, score)
                if update_flag:
                    if save_model:
                        self.save_model(self.saved_model_file, self.optimizer, self.scheduler, epoch_idx, self.cur_step, res, self.config)
                    self.best_valid_result = res
                else:
                    self.logger.info("No better score in the epoch. Patience: %d / %d", self.cur_step, self.early_stop)
                if stop_flag:
                    self.logger.info("Finished training, best eval result in epoch %d", epoch_idx - self.cur_step)
                    break
                if self.scheduler and epoch_idx > 0:
                    self.scheduler.step(score)
                    self.logger.info("epoch: %d, learning rate: %s", epoch_idx, self.optimizer.param_groups[0]["lr"])
            t0 = time.time()
            borrowed = None
            if hasattr(train_data, "use_stream") and hasattr(self.optimizer, "plan_stream"):
                # one stream for everything that runs a batch AHEAD; train_step(cur, nxt) plans `nxt` there and joins it (see use_stream).
                # The promise behind joined=True is kept by THIS loop only, so the loader gets its own stream and the waiting hand-out
                # back when the epoch ends, however it ends: evaluate(), a user loop or a train_step override that skips prefetch_plan
                # must never read a batch the plan stream has not built yet
                borrowed = (train_data.stream, train_data.joined)
                train_data.use_stream(self.optimizer.plan_stream(), joined=self.world == 1)
            draws = getattr(train_data, "loss_check_draws", None)
            if draws is not None and self.model.loss_type in ("bpr", "ccl"):
                train_data.loss_check_draws = 1     # (the reference's per-step random.random(): see BatchLoader)
            try:
                epoch_sum, n_nan = self._run_epoch(train_data)
            finally:
                if draws is not None:
                    train_data.loss_check_draws = draws
                if borrowed is not None:
                    train_data.stream, train_data.joined = borrowed
            if n_nan:
                self.logger.error("Training loss is nan in %d steps of epoch %d", n_nan, epoch_idx + 1)
            # reported train loss = SUM over batches of the batch loss (trainer.py:354-355)
            self.logger.info("epoch %d training [time: %.2fs, train loss: %.4f]", epoch_idx + 1, time.time() - t0, epoch_sum)
        return self.best_valid_score

    def _run_epoch(self, train_data):
        """the loop body of Trainer.fit (trainer.py:327-357) over one epoch -> (sum of the batch losses, number of NaN steps)"""
        losses, it = [], iter(train_data)
        epoch_sum, n_nan = 0.0, 0

        def drain():      # device scalars -> host: one sync per 4096 steps (and one per epoch), not one per step
            nonlocal epoch_sum, n_nan
            if not losses:
                return
            vals = np.asarray(torch.stack(losses).tolist(), dtype=np.float64)      # one launch, one copy
            losses.clear()
            nan = np.isnan(vals)
            n_nan += int(nan.sum())
            epoch_sum += float(vals[~nan].sum())      # (a Python-float running sum of loss.item(), as trainer.py:354-355)
            self.step_losses.extend(vals.tolist())

        cur = next(it, None)
        while cur is not None:          # one batch of lookahead: the next batch's plan overlaps this step
            nxt = next(it, None)
            losses.append(self.train_step(cur, nxt))
            if len(losses) >= 4096:
                drain()
            cur = nxt
        drain()
        return epoch_sum, n_nan

    # ------------------------------------------------------------------ evaluation
    @staticmethod
    def _metrics_from_rank(r, n_scores):
        """hit/ndcg/mrr/group_auc from the 0-based rank of the positive (unirec/facility/evaluation/onepos.py:100-175)."""
        r = np.maximum(np.asarray(r, dtype=np.float64), 0.0)   # a rank is a count: never below 0
        out = {"mrr": float(np.mean(1.0 / (r + 1))), "group_auc": float(np.mean((n_scores - 1 - r) / max(n_scores - 1, 1)))}
        for k in (1, 3, 5, 10, 20, 50, 100):
            out[f"hit@{k}"] = float(np.mean(r < k))
            out[f"ndcg@{k}"] = float(np.mean(np.where(r < k, 1.0 / np.log2(r + 2), 0.0)))
        return out

    def _history_csr(self, device):
        """Device CSR of ``set_user_history``'s table (object array user -> item ids, or a HistoryCSR), built once."""
        from ..data.rows import HistoryCSR
        uh = getattr(self, "user_history", None)
        if uh is None:
            return None, None
        if getattr(self, "_hist_src", None) is not uh:
            self._hist_csr = uh if isinstance(uh, HistoryCSR) else HistoryCSR(uh)
            self._hist_src = uh
        return self._hist_csr.to_device(device)

    @torch.no_grad()
    def evaluate_full_items(self, eval_data):
        """one_vs_all protocol: rank of the target among ALL items except item 0 and the user's history
        (Evaluator.evaluate_with_full_items, unirec/facility/evaluation/evaluator_abc.py:189-278).  One fused
        GEMM+count per batch on the device; the reference's random +-1e-8 tie-breaking noise is not reproduced."""
        from .. import ops
        model = self.model
        ranks = []
        local_hist = None
        for batch in eval_data:
            kw = {k: batch[k] for k in ("user_id", "item_seq", "item_seq_len") if k in batch}
            if self.world > 1:    # rows fetched through the row exchange, counts on every rank's own shard
                opt = self.optimizer
                target = batch["item_id"].reshape(batch["item_id"].shape[0], -1)[:, 0].contiguous()
                cb, restore = opt.compact_batch({k: v for k, v in batch.items() if k != "item_id"})
                try:
                    user_emb = model.forward_user_emb(**{k: cb[k] for k in kw}).contiguous().clone()
                finally:
                    restore()
                if local_hist is None:
                    local_hist = opt.local_history(*self._history_csr(user_emb.device))
                ranks.append(opt.full_item_ranks(user_emb, target, user_id=batch.get("user_id"), local_hist=local_hist))
                continue
            user_emb = model.forward_user_emb(**kw).contiguous()
            target = batch["item_id"].reshape(user_emb.shape[0], -1)[:, 0].contiguous()
            hp, hs = self._history_csr(user_emb.device)
            uid = batch.get("user_id")
            if uid is None:
                hp = hs = None
            rank, _ = ops.full_rank(user_emb, model.item_embedding.weight.data, target, user_id=uid, hist_ptr=hp, hist_sorted=hs,
                                    user_bias=model.user_bias.data if model.has_user_bias else None,
                                    item_bias=model.item_bias.data if model.has_item_bias else None, tau=model.tau)
            ranks.append(rank)
        r = self._gather_for_metrics([x.cpu().numpy() for x in ranks], eval_data)
        return self._metrics_from_rank(r, model.n_items)

    def _gather_for_metrics(self, per_batch, eval_data):
        """accelerator.gather_for_metrics (unirec/facility/evaluation/evaluator_abc.py): the per-batch results of all ranks in
        dataset order (rank r holds batches r, r + W, ...), cut to the dataset length (ranks that wrapped around re-ran the first
        batches)."""
        if self.world == 1:
            return np.concatenate(per_batch) if per_batch else np.zeros(0)
        from .. import pgroup as dist
        box = [None] * self.world
        dist.all_gather_object(box, per_batch, group=self.optimizer.xchg.cpu_group or self.optimizer.xchg.group)
        out = [box[r][k] for k in range(len(per_batch)) for r in range(self.world) if k < len(box[r])]
        flat = np.concatenate(out) if out else np.zeros(0)
        n_total = None
        ds = getattr(eval_data, "dataset", None)
        if ds is not None:
            n_total = len(ds)
        elif hasattr(eval_data, "pairs"):
            n_total = len(eval_data.pairs)
        return flat[:n_total] if n_total is not None else flat

    @torch.no_grad()
    def evaluate(self, eval_data, load_best_model=True, model_file=None, verbose=0, predict_only=False):
        # pending lazy zero-gradient steps belong to the weights that are in memory NOW: apply them before a checkpoint
        # replaces those weights (flushing afterwards would replay the old momentum on top of the loaded checkpoint)
        self.optimizer.flush()
        if load_best_model and os.path.exists(model_file or self.saved_model_file):
            self.load_model(model_file or self.saved_model_file)
        self.model.eval()
        if getattr(self, "eval_protocol", None) == "one_vs_all" and not predict_only:
            return self.evaluate_full_items(eval_data)
        ranks = []
        for batch in eval_data:   # one_vs_k: positive in column 0
            kw = {k: batch[k] for k in ("user_id", "item_id", "item_seq", "item_seq_len") if k in batch}
            if self.world > 1:    # the rows this batch looks up come through the row exchange (compact tables, ids re-indexed)
                cb, restore = self.optimizer.compact_batch(batch)
                try:
                    _, scores, _, _ = self.model(**{k: cb[k] for k in kw})
                    scores = scores.clone()
                finally:
                    restore()
            else:
                _, scores, _, _ = self.model(**kw)
            if predict_only:
                ranks.append(scores.cpu().numpy())
                continue
            if scores.dim() == 1:   # user-item-label rows: group_size consecutive rows are one ranking list (onepos.py:105-107)
                gs = int(self.config.get("group_size", -1) or -1)
                if gs <= 0:
                    raise ValueError("evaluating one score per row (user-item-label data) needs a positive group_size")
                scores = scores.view(-1, gs)
            ranks.append((scores[:, 1:] > scores[:, :1]).sum(1).cpu().numpy())   # 0-based rank of the positive
        if predict_only:
            return self._gather_for_metrics(ranks, eval_data)
        return self._metrics_from_rank(self._gather_for_metrics(ranks, eval_data), scores.shape[1])

    # ------------------------------------------------------------------ checkpoints (trainer.py:368-412)
    def save_model(self, filename, optimizer=None, scheduler=None, epoch=0, step=0, valid_result=None, config=None):
        self.optimizer.flush()
        if self.world > 1:
            # unwrap_model(model).state_dict() on the main process (trainer.py:389-398): the FULL tables under the reference's
            # names, streamed from the shards to rank 0 chunk by chunk; the other ranks only take part in the collective
            state_dict = self.optimizer.gather_state_dict()
            if self.rank != 0:
                return
        else:
            state_dict = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        os.makedirs(os.path.dirname(filename), exist_ok=True)
        torch.save({"config": {k: v for k, v in (config or self.config).items() if k != "device"}, "cur_epoch": epoch,
                    "cur_step": step, "best_valid_score": self.best_valid_score,
                    "state_dict": state_dict,
                    "optimizer": {"t": self.optimizer.t, "algo": self.optimizer.algo, "param_groups": self.optimizer.param_groups},
                    "scheduler": self.scheduler.state_dict() if self.scheduler is not None else None}, filename)

    def load_model(self, model_file):
        if self.world > 1:     # rank 0 reads the file; tables are dealt out row by row, the rest is broadcast
            ck = torch.load(model_file, map_location="cpu", weights_only=False) if self.rank == 0 else None
            self.optimizer.scatter_state_dict(ck["state_dict"] if ck is not None else None)
            return
        ck = torch.load(model_file, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ck["state_dict"], strict=False)
        self.model.check_views()
        # the loaded rows are current as of the optimizer's step counter: nothing is pending for them (the reference
        # loads the state_dict only and keeps stepping its optimizer state, trainer.py:400-412)
        self.optimizer.mark_tables_current()
