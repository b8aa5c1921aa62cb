"""Base estimator: batching, fit / predict / evaluate with the reference's signatures
(recman/tf/core/DeepModel.py:21-228), driving a recman_amd engine on the GPU.

Differences from the reference, all deliberate and documented:
  * the DataFrame is encoded ONCE per call and kept on the GPU (the reference re-encodes
    every mini-batch on the host, DeepModel.py:190-202 / inputs.py:53-58);
  * `len(y) // batch_size + 1` batches as in the reference (DeepModel.py:49,188), but the
    trailing EMPTY batch that appears when len % batch_size == 0 is skipped (the
    reference would feed it and turn every variable into NaN);
  * strict_reference=True reproduces two reference quirks: evaluation inside fit() runs
    with dropout active (DeepModel.py:103-111) and a new optimizer is built for every
    batch (xDeepFM.py:121-126); the default evaluates without dropout and keeps the
    optimizer state.

Multi-GPU (new: the reference is single-process): when the process is one rank of a torch.distributed
job (`torchrun --nproc-per-node N train.py`, backend "nccl" = RCCL; every rank constructs the same model
on the same data and calls the same methods), the embedding table is ROW-SHARDED over the ranks
(recman_amd/dist.py; hparams["table_sharding"] = "auto" | "row" | "none") and fit() is data parallel:
each global mini-batch is split over the ranks, every rank runs fwd+bwd on its part, owners update
their shard rows, the dense parameters follow the all-reduced gradient.  predict() / evaluate() run
every rank over all rows in step (the lookups are an exchange) and return the full result everywhere.
"""
import logging
from time import time

import numpy as np
import torch
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils import check_random_state

from .. import engine as eng
from ..optim import FusedDenseOptimizer, Optimizer, SparseTableOptimizer
from .inputs import DataInputs, FeatureDictionary

log = logging.getLogger(__name__)


class DeepModel(BaseEstimator, TransformerMixin):
    model = None  # "deepfm" | "dcn" | "xdeepfm"

    def __init__(self, feat_dict: FeatureDictionary, hparams: dict, metrics, epoch, batch_size=64,
                 random_seed=2019, task="classification", strict_reference=False, device="cuda"):
        assert task in ["classification", "regression"], (
            "target can be either 'classification' for classification task or 'regression' "
            "for regression task")  # DeepModel.py:31-34
        self.task = task
        self.feat_dict = feat_dict
        self.hparams = dict(hparams)
        self.epoch = epoch
        self.batch_size = batch_size
        self.random_seed = random_seed
        self.metrics = metrics
        self.strict_reference = strict_reference
        self.device = device
        self._engine = None
        self._opt = None
        self._shard, self._w = None, None  # (rank, world) of a row-sharded table; this rank's batch share

    # ------------------------------------------------------------------ engine
    def _build(self):
        if self._engine is not None:
            return self._engine
        fd = self.feat_dict
        fd.check_supported()
        spec = eng.FeatureSpec([f.name for f in fd.embedding_feats],
                               [f.feat_size for f in fd.embedding_feats],
                               [f.name for f in fd.dense_feats],
                               [f.name for f in fd.multi_val_csv_feats],
                               [f.name for f in fd.sparse_val_feats],
                               linear_names=self._linear_names())
        hp = dict(self.hparams)
        hp["strict_reference"] = self.strict_reference
        self._shard = self._dist_info()
        if self._shard is not None:
            return self._build_sharded(spec, hp)
        e = eng.ENGINES[self.model](spec, hp["embedding_size"], hp, task=self.task, device=self.device)
        eng.init_reference(e, self.random_seed)
        self._engine = e
        self._opt = Optimizer(hp.get("optimizer", "adam"), hp.get("learning_rate", 1e-3))
        # row-wise sparse step for the table.  embedding_l2_reg / linear_l2_reg (1e-5 in the reference's default
        # hyper-parameters, hparams/xDeepFM.py:22-23) would force a DENSE table gradient (layers.py:188-193): with
        # the row-wise step they are applied lazily - reg * row for the rows a batch touches (hparams["lazy_l2"],
        # default on whenever the row-wise optimizer is in use; strict_reference keeps the reference's dense term)
        has_l2 = bool(hp.get("embedding_l2_reg", 0.0) or hp.get("linear_l2_reg", 0.0))
        want = hp.get("sparse_optimizer", e.rows.numel() > (1 << 24))
        lazy = bool(hp.get("lazy_l2", not self.strict_reference))
        self._sparse_opt = None
        fits = e.rows.shape[1] >= e.D + 8 and 8 <= e.D <= 64  # room for the bias / linear moments in the row
        if want and (lazy or not has_l2) and fits and hp.get("optimizer", "adam") in ("adam", "adagrad", "gd", "sgd"):
            e.hp["lazy_l2"] = has_l2
            self._sparse_opt = SparseTableOptimizer(e, hp.get("optimizer", "adam"), hp.get("learning_rate", 1e-3),
                                                    l2_embedding=hp.get("embedding_l2_reg", 0.0),
                                                    l2_linear=hp.get("linear_l2_reg", 0.0))
            # ... and the dense parameters in one launch (the dict-walking Optimizer stays the path
            # for densified table gradients: small tables, strict l2 terms, FM bias dropout)
            self._dense_fused = FusedDenseOptimizer(e, hp.get("optimizer", "adam"), hp.get("learning_rate", 1e-3))
        return e

    # ------------------------------------------------------------ row-sharded table (multi-GPU)
    def _dist_info(self):
        """(rank, world) when the table is to be row-sharded: this process is one rank of a
        torch.distributed job with more than one rank (or hparams["table_sharding"] == "row")."""
        import torch.distributed as dist

        mode = self.hparams.get("table_sharding", "auto")
        if mode == "none":
            return None
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or mode == "row"):
            return dist.get_rank(), dist.get_world_size()
        if mode == "row":
            return 0, 1
        return None

    def _build_sharded(self, spec, hp):
        from .. import dist as rdist

        rank, world = self._shard
        name = hp.get("optimizer", "adam")
        dev = torch.device(self.device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        e = rdist.make_sharded_engine(self.model, spec, hp["embedding_size"], hp, dev, rank, world,
                                      capacity_factor=hp.get("exchange_capacity_factor"),  # None: exact split sizes
                                      micro_batches=int(hp.get("micro_batches", 1)))
        e.task = self.task
        eng.init_reference(e, self.random_seed)       # dense parameters: the same stream on every rank
        e.st.init_reference(spec.offsets(), spec.feat_sizes, self.random_seed)
        self._engine = e
        self._opt = self._sparse_opt = None
        self._shard_opt = e.optimizer(name, hp.get("learning_rate", 1e-3))
        return e

    def _local_part(self, s, t):
        """This rank's rows [a, b) of the global batch [s, t) and its weight B_local / B_global (None:
        fewer rows than ranks - every rank skips the batch)."""
        rank, world = self._shard
        n = t - s
        base, rem = divmod(n, world)
        if base == 0:
            return None
        a = s + rank * base + min(rank, rem)
        b = a + base + (1 if rank < rem else 0)
        return a, b, (b - a) / n

    def _any_rank(self, flag):
        """True on every rank when `flag` is true on ANY rank (one tiny all_reduce through the host)."""
        if self._shard is None or self._shard[1] == 1:
            return bool(flag)
        import torch.distributed as dist

        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            t = t.to(self._engine.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    def _fit_sharded_batch(self, idx, dense, yt, mv=None, masks=None, parts=None):
        """One training step of the row-sharded engine on this rank's part of the global batch.  Every choice
        that changes the NUMBER or the SIZES of the collectives is made rank-uniformly: `parts` = (smallest,
        largest) part of the global batch over the ranks (fit() knows the split; fit_on_batch agrees on it with
        one all_reduce) decides whether the micro-batch pipeline runs (every part divisible by M) and sizes the
        fixed-capacity buckets (the largest part); a fixed-capacity batch that overflowed on ANY rank is redone
        with exact split sizes before the optimizer step."""
        e = self._engine
        M = e.micro_batches
        B = idx.shape[0]
        if parts is None:  # fit_on_batch: agree on the part sizes
            parts = (B, B)
            if self._shard[1] > 1:
                import torch.distributed as dist

                t = torch.tensor([-B, B], dtype=torch.int64)
                if dist.get_backend() == "nccl":
                    t = t.to(e.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                parts = (-int(t[0]), int(t[1]))
        lo, hi = parts
        ragged = lo != hi or lo % M != 0
        st = e.st
        fixed = st.capacity_factor
        if fixed and mv and e.spec.multi_names and not getattr(self, "_mv_caps_global", False):
            # the padded tag columns of the fixed-capacity layout need ONE width per multi-valued feature on every
            # rank: the most tags any example of this batch carries on any rank (fit() knows the whole dataset and
            # sets it once; here - fit_on_batch - it can only grow)
            names = list(e.spec.multi_names)
            t = torch.tensor([int((mv[n][0][1:] - mv[n][0][:-1]).max()) if mv[n][1].numel() else 0 for n in names],
                             dtype=torch.int64)
            if self._shard[1] > 1:
                import torch.distributed as dist

                if dist.get_backend() == "nccl":
                    t = t.to(e.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            caps = dict(e._mv_T)
            for n, v in zip(names, t.tolist()):
                caps[n] = max(int(caps.get(n, 0)), int(v), 1)
            e.set_mv_capacity(caps)
        redone = False
        try:
            if ragged or masks is not None:  # (dropout masks: no micro-batch pipeline, dist.py)
                e.micro_batches = 1
            if fixed:
                st.cap_occurrences = hi * e.exchange_columns()  # the same bucket size on every rank
            loss = e.fwd_bwd(idx, dense, yt, masks=masks, weight=self._w, mv=mv)
            if fixed and self._any_rank(e.overflowed()):
                # skewed ids: occurrences were clamped onto the last slot of a bucket - this step's rows and
                # gradients are wrong on some rank; redo it with exact split sizes (every rank, same decision)
                st.capacity_factor = None
                e._B = None
                redone = True
                loss = e.fwd_bwd(idx, dense, yt, masks=masks, weight=self._w, mv=mv)
        finally:
            e.micro_batches = M
            if fixed:
                st.capacity_factor, st.cap_occurrences = fixed, None
                if redone:
                    e._B = None  # (the buffers are sized for the layout: re-made for the next fixed-capacity batch)
        self._shard_opt.step(reset=self.strict_reference)
        return loss

    @property
    def variables(self):
        """name -> tensor with the reference's variable names (DeepModel.py:43)."""
        return self._build().state_dict()

    def _encode(self, X, y=None, on_host=False):
        inp = DataInputs().load(self.feat_dict, X, y)
        dev = "cpu" if on_host else self._build().device
        idx = torch.from_numpy(np.ascontiguousarray(inp.idx)).to(dev)
        dense = torch.from_numpy(np.ascontiguousarray(inp.dense)).to(dev)
        yt = None
        if y is not None:
            ya = np.asarray(y)
            yt = torch.from_numpy(ya.astype(np.int64 if self.task == "classification" else np.float32)).to(dev)
        self._mv_host = inp.mv  # name -> CSR (host) of the multi-valued features of the LAST encoded frame
        return idx, dense, yt

    def _mv_batch(self, mv_host, s, t):
        """Device (offsets, ids) of examples [s, t) for every multi-valued feature (or None)."""
        if not mv_host:
            return None
        dev = self._build().device
        out = {}
        for name, csr in mv_host.items():
            c = csr.slice(s, t)
            out[name] = (torch.from_numpy(c.offsets).to(dev), torch.from_numpy(c.ids).to(dev))
            if c.vals is not None:  # SparseValueFeat: (offsets, ids, vals)
                out[name] += (torch.from_numpy(c.vals).to(dev),)
        return out

    def _linear_names(self):
        """The hyper-parameter `linear_features` (get_linear_features, utils.py:27-30): a
        comma-separated list of feature names = the linear term's features in that order; empty =
        all of them.  Returns the list or None."""
        lf = (self.hparams or {}).get("linear_features")
        if not lf:
            return None
        names = [n.strip() for n in lf.split(",")] if isinstance(lf, str) else list(lf)
        for n in names:
            if n not in self.feat_dict:
                raise KeyError(f"linear_features: no feature named {n!r}")  # feat_dict[name] in the reference
        return names

    def _manual_weights(self):
        """Concatenated per-feature manual weights in linear-feature order
        (layers.py:338-345), or None when no feature has any."""
        fd = self.feat_dict
        names = self._linear_names()
        # utils.py:27-36: the named features in the given order, or sparse, value, multi-valued csv, dense
        feats = [fd[n] for n in names] if names else fd.linear_feats
        if not any(getattr(f, "_weights", None) for f in feats):
            return None
        w = np.concatenate([np.asarray(f.weights, dtype=np.float64).reshape(-1) for f in feats])
        return torch.from_numpy(w.astype(np.float32))

    # ------------------------------------------------------------------ predict
    @staticmethod
    def get_batch(X, y, batch_size, index):
        start = index * batch_size
        end = min(start + batch_size, len(y))
        return X[start:end], y[start:end]

    def _predict_encoded(self, idx, dense, training, mv_host=None):
        e = self._build()
        n = idx.shape[0]
        out = np.empty((n,), dtype=np.float32)
        mw = None if training else self._manual_weights()
        if mw is not None and self._shard is not None:
            raise NotImplementedError("row-sharded table: manual feature weights are single-GPU only")
        total_batch = n // self.batch_size + 1  # DeepModel.py:49
        for bi in range(total_batch):
            s, t = bi * self.batch_size, min((bi + 1) * self.batch_size, n)
            if t <= s:
                continue
            masks = self._dropout_masks(t - s) if training else None
            ib, db = idx[s:t].contiguous(), dense[s:t].contiguous()
            if not ib.is_cuda:  # dataset kept on the host (feeder="pinned")
                ib, db = ib.to(e.device, non_blocking=True), db.to(e.device, non_blocking=True)
            _, pred = e.forward(ib, db, training=training,
                                masks=masks, manual_weights=mw, mv=self._mv_batch(mv_host, s, t))
            out[s:t] = pred.cpu().numpy()
        return out

    def predict(self, X, training=False, batch_number_to_show_progress=50):
        idx, dense, _ = self._encode(X)
        return self._predict_encoded(idx, dense, training, self._mv_host)

    def evaluate(self, X, y, training=False, batch_number_to_show_progress=50):
        pred = self.predict(X, training, batch_number_to_show_progress)
        return [metric(y, pred) for metric in self.metrics]

    # ---------------------------------------------------------------------- fit
    def _dropout_masks(self, B):
        """0/1 keep masks for the configured keep-probabilities (tf.nn.dropout,
        layers.py:461,466,589,602), drawn on the GPU."""
        e = self._build()
        hp, dev = e.hp, e.device
        masks = {}
        n = len(hp.get("deep_hidden_units", ()))
        keep = hp.get("deep_dropout")
        if keep is not None and any(k < 1 for k in keep) and getattr(e, "mlp", None) is not None:
            dims = [e.FD + e.Dn] + list(hp["deep_hidden_units"])
            masks["dnn"] = [(torch.rand(B, d, device=dev) < k).float() if k < 1 else None
                            for d, k in zip(dims, keep)]
        fk = hp.get("fm_dropout")
        if fk is not None and any(k < 1 for k in fk) and e.model == "deepfm":
            mb = (torch.rand(B, e.F, device=dev) < fk[0]).float() / fk[0] if fk[0] < 1 else None
            me = (torch.rand(B, e.F, e.D, device=dev) < fk[1]).float() / fk[1] if fk[1] < 1 else None
            masks["fm"] = (mb, me)
        ck = hp.get("cin_dropout")
        if ck is not None and any(k < 1 for k in ck) and e.model == "xdeepfm":
            shapes = [(B, e.F, e.D)] + [(B, n, e.D) for n in e.units]
            masks["cin"] = [(torch.rand(*sh, device=dev) < k).float() if k < 1 else None
                            for sh, k in zip(shapes, ck)]
        return masks or None

    def fit_on_batch(self, X, y):
        idx, dense, yt = self._encode(X, y)
        return self._fit_encoded(idx, dense, yt, self._mv_batch(self._mv_host, 0, idx.shape[0]))

    def _fit_encoded(self, idx, dense, yt, mv=None):
        e = self._build()
        if idx.shape[0] == 0:
            return None
        if self._shard is not None:
            self._w = getattr(self, "_w", None)
            parts, self._parts = getattr(self, "_parts", None), None
            return self._fit_sharded_batch(idx, dense, yt, mv, masks=self._dropout_masks(idx.shape[0]), parts=parts)
        side = None
        if self._sparse_opt is not None and not e.spec.scratch_names:
            # the id-only third of the row-wise step (keys + sort by row) runs on a side stream beside the
            # forward+backward pass
            self._sparse_opt._workspace(idx.numel())  # (allocated on the main stream)
            side = self._side_stream = getattr(self, "_side_stream", None) or torch.cuda.Stream(device=e.device)
            side.wait_stream(torch.cuda.current_stream(e.device))
            with torch.cuda.stream(side):
                self._sparse_opt.prepare(idx)
        loss = e.fwd_bwd(idx, dense, yt, masks=self._dropout_masks(idx.shape[0]), mv=mv)
        if side is not None:
            torch.cuda.current_stream(e.device).wait_stream(side)
        if self.strict_reference:
            self._opt.reset()  # a NEW optimizer every batch (xDeepFM.py:121-126)
        fm_masked = getattr(e, "d_bias", None) is not None  # FM bias dropout: per-occurrence grads
        if self._sparse_opt is not None and not fm_masked:
            self._sparse_opt.step(idx, reset=self.strict_reference)
            self._dense_fused.step(reset=self.strict_reference)  # dense parameters only
        else:
            self._opt.step(e.params, e.dense_grads(idx))
        return loss

    def _eval_at_epoch(self, enc_train, y_train, enc_valid=None, y_valid=None, start_time=None, epoch=0):
        training = bool(self.strict_reference)  # DeepModel.py:103-111 evaluates with training=True
        ptr = self._predict_encoded(enc_train[0], enc_train[1], training, enc_train[2])
        tr = [m(y_train, ptr) for m in self.metrics]
        va = None
        if enc_valid is not None:
            pva = self._predict_encoded(enc_valid[0], enc_valid[1], training, enc_valid[2])
            va = [m(y_valid, pva) for m in self.metrics]
        log.info("[%d] train-result=%s%s [%.1f s]", epoch, [round(float(r), 4) for r in tr],
                 "" if va is None else ", valid-result=%s" % [round(float(r), 4) for r in va],
                 time() - (start_time or time()))
        return tr, va

    def fit(self, X_train, y_train, X_valid=None, y_valid=None, random_seed_for_mini_batch=True,
            tb_logger=None, epoch_callback=None, show_progress=False,
            batch_number_to_show_progress=50):
        assert X_train is not None or y_train is not None  # DeepModel.py:153
        y_train = np.asarray(y_train)
        enc_valid = None
        if X_valid is not None and y_valid is not None:
            vi, vd, _ = self._encode(X_valid)
            enc_valid = (vi, vd, self._mv_host)
        pinned = self._use_feeder(len(y_train))
        idx, dense, yt = self._encode(X_train, y_train, on_host=pinned)
        mv_host = self._mv_host
        if self._shard is not None and mv_host and getattr(self._engine.st, "capacity_factor", None):
            # every rank encodes the whole frame: the widest tag list of the dataset, the same number everywhere
            self._engine.set_mv_capacity({n: max(1, int(np.diff(c.offsets).max()) if len(c.ids) else 1)
                                          for n, c in mv_host.items() if n in self._engine.spec.multi_names})
            self._mv_caps_global = True
        n = len(y_train)
        if pinned:
            return self._fit_pinned(idx, dense, yt, mv_host, y_train, enc_valid, y_valid,
                                    random_seed_for_mini_batch, epoch_callback, X_train,
                                    batch_number_to_show_progress)
        eval_results = self._eval_at_epoch((idx, dense, mv_host), y_train, enc_valid, y_valid, time())
        for epoch in range(1, self.epoch + 1):
            start = time()
            seed = np.random.randint(1, 2019) if random_seed_for_mini_batch else self.random_seed
            seed = self._same_on_all_ranks(seed)
            # sklearn.utils.shuffle(X, random_state=seed) (DeepModel.py:182-187): the same
            # permutation, applied to the encoded arrays instead of the DataFrame
            perm = np.arange(n)
            check_random_state(seed).shuffle(perm)
            pt = torch.from_numpy(perm).to(idx.device)
            idx, dense, yt = idx[pt], dense[pt], yt[pt]
            y_train = y_train[perm]
            mv_host = {k: c.take(perm) for k, c in mv_host.items()} if mv_host else mv_host
            total_batch = n // self.batch_size + 1  # DeepModel.py:188
            for i in range(total_batch):
                s, t = i * self.batch_size, min((i + 1) * self.batch_size, n)
                if t <= s:
                    continue  # the reference's trailing empty batch
                if self._shard is not None:  # data parallel: this rank's part of the global batch
                    part = self._local_part(s, t)
                    if part is None:
                        continue
                    base, rem = divmod(t - s, self._shard[1])
                    self._parts = (base, base + (1 if rem else 0))  # smallest / largest part over the ranks
                    s, t, self._w = part
                self._fit_encoded(idx[s:t].contiguous(), dense[s:t].contiguous(), yt[s:t].contiguous(),
                                  self._mv_batch(mv_host, s, t))
                if i % batch_number_to_show_progress == 0:
                    log.info(f"Fit: {(i + 1)}/{total_batch} has been completed")
            eval_results = self._eval_at_epoch((idx, dense, mv_host), y_train, enc_valid, y_valid, start, epoch)
            if epoch_callback:
                epoch_callback(model=self, eval_results=eval_results, df_all=X_train[:1])
        return None  # the reference's fit returns None

    def _same_on_all_ranks(self, value):
        """Rank 0's value on every rank (the per-epoch shuffle seed must agree)."""
        if self._shard is None or self._shard[1] == 1:
            return value
        import torch.distributed as dist

        box = [value]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def _use_feeder(self, n):
        if self._build() is not None and self._shard is not None:
            return False  # (sharded fit keeps the encoded dataset on the GPU)
        """hparams["feeder"]: "gpu" (whole encoded dataset resident in HBM, the default while it is
        small), "pinned" (host-resident, batches through the pinned-memory feeder), or "auto":
        pinned once the encoded arrays exceed a quarter of the free HBM."""
        mode = self.hparams.get("feeder", "auto")
        if mode == "gpu":
            return False
        if mode == "pinned":
            return True
        from .feeder import encoded_nbytes

        e = self._build()
        free, _ = torch.cuda.mem_get_info(e.device)
        return encoded_nbytes(n, e.F, e.Dn) > free // 4

    def _fit_pinned(self, idx, dense, yt, mv_host, y_train, enc_valid, y_valid, random_seed_for_mini_batch,
                    epoch_callback, X_train, every):
        """fit() with the encoded dataset in pinned host memory (th/feeder.py): same batches, same
        shuffles, the H2D copy of batch i+1 overlaps the step on batch i."""
        from .feeder import BatchFeeder

        e = self._build()
        n = len(y_train)
        feeder = BatchFeeder(idx, dense, yt, self.batch_size, e.device)
        perm_all = np.arange(n)  # position -> original row: the shuffles compose across epochs
        eval_results = self._eval_at_epoch((idx, dense, mv_host), y_train, enc_valid, y_valid, time())
        for epoch in range(1, self.epoch + 1):
            start = time()
            seed = np.random.randint(1, 2019) if random_seed_for_mini_batch else self.random_seed
            p = np.arange(n)
            check_random_state(seed).shuffle(p)
            perm_all = perm_all[p]
            mv_perm = {k: c.take(perm_all) for k, c in mv_host.items()} if mv_host else mv_host
            total_batch = n // self.batch_size + 1
            for i, (s, t, ib, db, yb) in enumerate(feeder.batches(perm_all)):
                self._fit_encoded(ib, db, yb, self._mv_batch(mv_perm, s, t))
                if i % every == 0:
                    log.info(f"Fit: {(i + 1)}/{total_batch} has been completed")
            # evaluation order does not matter for the metrics: the unshuffled host arrays
            eval_results = self._eval_at_epoch((idx, dense, mv_host), y_train, enc_valid, y_valid, start, epoch)
            if epoch_callback:
                epoch_callback(model=self, eval_results=eval_results, df_all=X_train[:1])
        return None

    # --------------------------------------------------------------- checkpoint
    @classmethod
    def from_hparams(cls, feat_dict, hparams, **ctor):
        """Builds the model from a saved hparams dict (BestModelFinder.load): gen-2 classes take
        the dict itself (xDeepFM.py:26-35), gen-1 classes keyword arguments (DeepFM.py:30-53)."""
        import inspect

        names = inspect.signature(cls.__init__).parameters
        if "hparams" in names:
            return cls(feat_dict, dict(hparams), **ctor)
        return cls(feat_dict, **{k: v for k, v in hparams.items() if k in names and k not in ctor}, **ctor)

    def save(self, path):
        """state_dict with the reference's variable names (tf.train.Checkpoint(**variables),
        BestModelFinder.py:57-68)."""
        e = self._build()
        if self._shard is not None:
            # one file per rank for its shard rows (<path>.shard<r>of<W>.pt) + the dense parameters (rank 0)
            e.st.save(path)
            if self._shard[0] == 0:
                torch.save({k: v.detach().cpu() for k, v in e.params.items() if k != "table_shard"}, path)
            return
        torch.save({k: v.cpu() for k, v in e.state_dict().items()}, path)

    def restore(self, path="ckpt_model.pt"):
        e = self._build()
        if self._shard is not None:
            if self._shard[1] > 1:
                import torch.distributed as dist

                dist.barrier()  # rank 0's file must be complete
            for k, v in torch.load(path, weights_only=True).items():
                e.params[k].copy_(v.to(e.device))
            e.st.load(path)
            return
        e.load_params(torch.load(path, weights_only=True))
