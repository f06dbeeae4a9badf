"""The training step of the reference on the HIP kernels.

Mirrors bin/train_network.py:124-142: `wrap_network(network, min_prob, l2, drop)` returns `fg(x, labels, weights, rate)
-> (loss, acc)` which also applies one ADAMski update (sloika/updates.py:36-89) to the network's parameters, exactly
like the Theano function the reference compiles.  Device side: csrc/train.hip (reverse scan of the GRU layers, A^T B
contractions for the weight gradients, softmax cross-entropy, the optimiser) on top of the inference kernels, which
ARE the forward pass.  There is no CPU fallback.

Supported networks: a Serial that ends in Softmax and is built from Convolution (any number of input features, anywhere but
inside a Reverse), Window as the first layer, Gru (up to 144 wide), Lstm (up to 128 wide; odd widths run zero-padded),
FeedForward, Reverse, Parallel (so `birnn`) and nested Serial -- the raw-signal models models/raw_0.98_rgrgr.py,
baseline_raw_gru.py, bigger_raw_gru.py, raw_1.00_rGr.py (its 110/142-wide layers run zero-padded) and the event-feature
models baseline_gru.py / tiny_gru.py / baseline_lstm.py; anything else (wider recurrent layers, other activations) raises
NotImplementedError: the reference differentiates any layer through Theano, only the raw-signal GRU path is accelerated here.

Data parallel (BASELINE.json configs[4]): with torch.distributed initialised (backend "nccl" = RCCL over xGMI) every
rank runs the same step on its own chunks, the flat float32 gradient is summed with ONE all-reduce and divided by the
world size, and every rank applies the identical update -- the loss of the global batch is the mean of the ranks'
losses because every rank counts the same number of positions.
"""
import numpy as np

from . import _lib, activation, layers, profiler
from .config import sloika_dtype


def remove_blanks(labels):
    """Non-transducer labels (train_network.py:116-121): every blank (0) after the first position takes the label that
    precedes it, in place, row by row.  Vectorised: carry the index of the last non-blank position forward."""
    labels = np.asarray(labels)
    for row in labels:
        pos = np.where(row != 0, np.arange(len(row)), 0)
        pos[0] = 0                                            # position 0 keeps whatever it holds, blank included
        row[:] = row[np.maximum.accumulate(pos)]
    return labels


class ExponentialSmoother(object):
    """Exponentially weighted running mean with bias correction, as used for the progress lines
    (train_network.py:100-113): value = sum_k f^(n-k) (1-f) v_k / sum_k f^(n-k) (1-f) w_k."""

    def __init__(self, factor, val=0.0, weight=1e-30):
        if not 0.0 <= factor <= 1.0:
            raise AssertionError("Smoothing factor was {}, should be between 0.0 and 1.0.\n".format(factor))
        self.factor, self.val, self.weight = factor, val, weight

    @property
    def value(self):
        return self.val / self.weight

    def update(self, val, weight=1.0):
        keep, take = self.factor, 1.0 - self.factor
        self.val = keep * self.val + take * val
        self.weight = keep * self.weight + take * weight


def adamski_scalars(t, rate, decay, mrate=0.0005):
    """(lr_t, momentum_decay, t + 1) of the step that starts at counter `t` (updates.py:54-76), in float32 like the
    reference's shared variables."""
    f32 = np.float32
    if mrate is not None:
        m_rate = -f32(mrate)
        m_p = np.exp(m_rate)
        m_k = f32((1.0 - decay[0]) * decay[0] * m_p / (1.0 - m_p * decay[0]))
    else:
        m_rate, m_k = -f32(1e30), f32(0.0)
    ldecay = np.log(np.array(decay), dtype=np.float32)
    t = f32(t)
    t_new = f32(t + f32(1.0))
    with np.errstate(over="ignore"):
        momentum_factor = f32(m_k * np.expm1(f32(t * f32(ldecay[0] + m_rate))) - np.expm1(f32(t_new * ldecay[0])))
        lr_t = f32(f32(rate) * np.sqrt(-np.expm1(f32(t_new * ldecay[1]))) / momentum_factor)
        momentum_decay = f32(-f32(decay[0]) * np.expm1(f32(t_new * m_rate)))
    return float(lr_t), float(momentum_decay), float(t_new)


def allreduce_mean_(tensor):
    """Sum `tensor` over the ranks in place (one collective) and return the factor that turns the sum into the mean.
    A no-op returning 1.0 when torch.distributed is not initialised or there is one rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1.0
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


def allreduce_step_scalars_(sc):
    """The scalars a rank reads after a training step, [loss sum, accuracy sum, sum of squared parameters, bad-label flag]: the two
    sums and the FLAG are summed over the ranks in one collective (the parameter term is the same on every rank and stays), so that a
    batch with a label out of range on ANY rank makes EVERY rank raise before its update -- the rank with the bad label has sent a
    garbage gradient into the all-reduce, and a flag kept per rank would let the others apply it.  A no-op with one rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sc
    idx = torch.tensor([0, 1, 3], device=sc.device)
    part = sc.index_select(0, idx)
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    sc.index_copy_(0, idx, part)
    return sc


def rank_and_world():
    """(rank, world size) of the initialised process group, (0, 1) without one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_from_rank0_(tensor):
    """Every rank takes rank 0's values (in place).  A no-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=0)
    return tensor


def _dx_gemm(a_ptr, lda, wt, out_ptr, ldo, M, what, below=None):
    """out[M][N] = a[M][K] . wt[N][K]^T, the dL/dx product of a layer (wt = its transposed weight, contiguous): six bf16 MFMA terms
    per product (csrc/gemm_bf16x6.hip) unless SLOIKA_AMD_EXACT_F32=1 or the shape is not covered, then the fp32 matrix pipe.
    `below` = (y, activation id) of an element-wise activation whose output is this layer's input: the product then leaves multiplied
    by fun'(.) -- it IS dL/d(pre-activation) of the layer below (slk_gemm_dact_bf16x6) -- and True is returned; False when that
    fusion did not happen (the caller applies slk_act_backward_f32 as before)."""
    import torch
    L = _lib.lib()
    N, K = int(wt.shape[0]), int(wt.shape[1])
    rc, fused = _lib.SLK_ERR_UNSUPPORTED, False
    if layers.SPLIT_F16 and K % 4 == 0 and lda % 4 == 0 and K >= 32:
        packed = torch.empty(L.slk_pack_bf16x3_bytes(N, K), dtype=torch.uint8, device=wt.device)
        _lib.check(L.slk_pack_bf16x3_f32(wt.data_ptr(), N, K, packed.data_ptr(), layers._stream()), what)
        if below is not None:
            yb, act = below
            rc = L.slk_gemm_dact_bf16x6(a_ptr, lda, packed.data_ptr(), yb.data_ptr(), layers._row_stride(yb), act, out_ptr, ldo, M, K, N,
                                        layers._stream())
            fused = rc == _lib.SLK_OK
        if not fused:
            rc = L.slk_gemm_bias_act_bf16x6(a_ptr, lda, packed.data_ptr(), None, out_ptr, ldo, M, K, N, 0, layers._stream())
    if rc == _lib.SLK_ERR_UNSUPPORTED:
        rc = L.slk_gemm_bias_act_f32(a_ptr, lda, wt.data_ptr(), None, out_ptr, ldo, M, K, N, 0, layers._stream())
    _lib.check(rc, what)
    return fused


def _unwrap(layer, rev=False):
    while isinstance(layer, layers.Reverse):
        layer, rev = layer.layer, not rev
    return layer, rev


#: the softmax layer's loss gradient from two passes over its products, no logits in memory (csrc/gemm_rows_f16x3.hip)
XENT_TWO_PASS = "xent_in_place" not in layers._DEBUG
_FF_ACTS = ("linear", "tanh", "sigmoid", "relu", "elu")     # activations whose derivative is a function of the output


def _validate(layer, first, rev=False, where="network"):
    """Raise NotImplementedError unless `layer` (a sub-tree in front of the Softmax) can be differentiated here."""
    layer, rev = _unwrap(layer, rev)
    name = type(layer).__name__
    if isinstance(layer, layers.Serial):
        for k, sub in enumerate(layer.layers):
            _validate(sub, first and k == 0, rev, where)
    elif isinstance(layer, layers.Parallel):
        for sub in layer.layers:
            _validate(sub, False, rev, where)
    elif isinstance(layer, layers.Convolution):
        if rev:
            raise NotImplementedError("training: Convolution inside a Reverse is not covered")
        if activation.act_name(layer.fun) not in _FF_ACTS:
            raise NotImplementedError("training: Convolution activation %s has no derivative kernel" % layer.fun.__name__)
    elif isinstance(layer, layers.Window):
        if not first:
            raise NotImplementedError("training: Window only as the first layer (the event-feature front end), where it needs "
                                      "no reverse pass")
    elif isinstance(layer, layers.Gru):
        if activation.act_name(layer.fun) != "tanh" or activation.act_name(layer.gatefun) != "sigmoid":
            raise NotImplementedError("training: Gru layers with fun=tanh, gatefun=sigmoid only")
        if (layer.size + 15) // 16 * 16 not in (16, 32, 48, 64, 96, 112, 128, 144):
            raise NotImplementedError("training: no reverse-scan kernel for a Gru of size %d" % layer.size)
    elif isinstance(layer, layers.Lstm):
        if activation.act_name(layer.fun) != "tanh" or activation.act_name(layer.gatefun) != "sigmoid":
            raise NotImplementedError("training: Lstm layers with fun=tanh, gatefun=sigmoid only")
        if layer.size > 128:
            raise NotImplementedError("training: no reverse-scan kernel for an Lstm of size %d" % layer.size)
    elif isinstance(layer, layers.FeedForward):
        if activation.act_name(layer.fun) not in _FF_ACTS:
            raise NotImplementedError("training: FeedForward activation %s has no derivative kernel" % layer.fun.__name__)
    else:
        raise NotImplementedError(
            "training on the GPU path covers Convolution, Window first, Gru, Lstm, FeedForward, Reverse, Parallel, "
            "Serial and a final Softmax; %s (%s) is outside it" % (name, where))


def _plan(network):
    """(body, softmax): the layers in front of the output layer as a Serial-like list, and the Softmax; raises
    NotImplementedError for anything the reverse pass does not cover."""
    subs = list(network.layers) if isinstance(network, layers.Serial) else [network]
    if not subs or not isinstance(subs[-1], layers.Softmax):
        raise NotImplementedError("the training loss needs a Softmax output layer (train_network.py:128-133)")
    for k, sub in enumerate(subs[:-1]):
        _validate(sub, k == 0, where="layer %d" % k)
    return subs[:-1], subs[-1]


def _leaves(layer):
    layer, _ = _unwrap(layer)
    if isinstance(layer, (layers.Serial, layers.Parallel)):
        return [leaf for sub in layer.layers for leaf in _leaves(sub)]
    return [layer]


class TrainingStep(object):
    """fg(x, labels, weights, rate) -> (loss, acc): one forward/backward pass and one optimiser update.

    x       : [T, B, nfeature] float32 (numpy or device tensor), time-major like train_network.py:304
    labels  : [T', B] int32, T' = the network's output length (train_network.py:305)
    weights : [T', B] float32 (train_network.py:306)
    rate    : learning rate of this step (train_network.py:289)
    """

    def __init__(self, network, min_prob=0.0, l2=0.0, drop=0, decay=(0.9, 0.999), epsilon=1e-8, clip=5.0, mrate=0.0005,
                 optimiser="adam", momentum=0.9):
        import torch
        _lib.require_gpu()
        assert 0.0 < decay[0] < 1.0 and 0.0 < decay[1] < 1.0, "Decay must lie strictly between zero and one"   # updates.py:51-52
        assert mrate is None or mrate > 0.0, "Rate of momentum increase must be positive"                      # updates.py:53
        assert optimiser in ("adam", "sgd")
        self.network = network
        self.body, self.softmax = _plan(network)
        self.min_prob, self.l2, self.drop = float(min_prob), float(l2), int(drop)
        self.decay, self.epsilon, self.clip, self.mrate = tuple(decay), float(epsilon), float(clip), mrate
        self.optimiser, self.sgd_momentum = optimiser, float(momentum)
        self.t = 0.0
        from . import device as D
        dev = D.device()
        # one flat parameter buffer; every Shared's device mirror becomes a view into it, so the forward kernels read the
        # optimiser's output directly and the gradient can be all-reduced as one message
        self.shared = network.params()
        sizes = [int(np.prod(p.shape)) for p in self.shared]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        total = int(self.offsets[-1])
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        for p, off, n in zip(self.shared, self.offsets, sizes):
            self.flat[off:off + n].copy_(torch.from_numpy(p.get_value(borrow=True).reshape(-1)))
            p._dev = self.flat[off:off + n].view(p.shape)
            p._device_is_master = True                 # get_value() / pickling now read the optimiser's buffer
        broadcast_from_rank0_(self.flat)                # data-parallel replicas start from rank 0's parameters
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.momentum = torch.zeros(total, dtype=torch.float32, device=dev)
        self.variance = torch.zeros(total, dtype=torch.float32, device=dev) if optimiser == "adam" else None
        self._index = {id(p): i for i, p in enumerate(self.shared)}
        self._ws = None
        self._scalars = torch.zeros(4, dtype=torch.float64, device=dev)     # loss sum, accuracy sum, sum of squared parameters, bad-label flag
        self._sum_scratch = torch.empty(2 * 256, dtype=torch.float64, device=dev)
        self._below, self._dpre_ready, self._dy_is_dpre = None, False, False
        self._drop_caches()

    # ---- plumbing -------------------------------------------------------------------------------------------------
    def _grad_of(self, shared_param):
        i = self._index[id(shared_param)]
        return self.grad[self.offsets[i]:self.offsets[i + 1]]

    def _drop_caches(self):
        """Device copies derived from the parameters (fp16 splits, padded twins) are keyed on the identity of the
        parameter's device tensor, which no longer changes when the optimiser writes in place: forget them."""
        for p in self.shared:                            # every cache keys on (device tensor, version): a new version ...
            p._version = getattr(p, "_version", 0) + 1
        for layer in [leaf for sub in self.body for leaf in _leaves(sub)] + [self.softmax]:
            for attr in ("_w16", "_wbf16", "_iw16", "_pad_cache", "_svpack", "_svpack_p"):    # ... and the old tensors are released right away
                layer.__dict__.pop(attr, None)
            # (a Gru whose forward pass runs a zero-padded twin rebuilds it from get_value(), which reads the device copy)

    def sync_host(self):
        """Copy the trained parameters back into the layers' numpy storage (what pickling a network saves:
        train_network.py:145-152)."""
        host = self.flat.cpu().numpy()
        for p, off, nxt in zip(self.shared, self.offsets[:-1], self.offsets[1:]):
            p._value = np.ascontiguousarray(host[off:nxt].reshape(p.shape), dtype=sloika_dtype)

    def _workspace(self, nbytes):
        import torch
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.flat.device)
        return self._ws

    def _tn(self, A, lda, Bm, ldb, C, ldc, M, n1, n2, colsum=None):
        """C[n1][n2] = A^T B over M rows; colsum (optional) = A^T 1."""
        L = _lib.lib()
        nbytes = L.slk_gemm_tn_workspace_bytes(M, n1, n2)
        ws = self._workspace(nbytes)
        # weight gradients on the bf16 pipe as six-term splits (float32-grade; measured 3-8 % faster than the fp32 MFMA form,
        # both are bound by their dword loads) unless plain fp32 MFMA is asked for or the result is a sliver
        fn = L.slk_gemm_tn_bf16x6_f32 if (layers.SPLIT_F16 and n2 >= 32) else L.slk_gemm_tn_f32
        _lib.check(fn(A, lda, Bm, ldb, C, ldc, M, n1, n2, colsum, ws.data_ptr(), nbytes, layers._stream()), "gemm_tn")

    def _tn_many(self, problems, M):
        """Several contractions over the same M rows -- problems = [(A, lda, B, ldb, C, ldc, n1, n2, colsum or None), ...] with device
        addresses -- in one launch where the bf16 kernel applies (slk_gemm_tn_multi_bf16x6_f32: a matrix that several problems
        share is read from memory once), else one by one."""
        import ctypes
        L = _lib.lib()
        n = len(problems)
        if layers.SPLIT_F16 and 1 < n <= 4 and all(p[7] >= 32 for p in problems):
            vps, longs, ints = (ctypes.c_void_p * n), (ctypes.c_long * n), (ctypes.c_int * n)
            A, lda, Bm, ldb = vps(*[p[0] for p in problems]), longs(*[p[1] for p in problems]), vps(*[p[2] for p in problems]), longs(*[p[3] for p in problems])
            C, ldc = vps(*[p[4] for p in problems]), longs(*[p[5] for p in problems])
            n1, n2, cs = ints(*[p[6] for p in problems]), ints(*[p[7] for p in problems]), vps(*[p[8] for p in problems])
            nbytes = L.slk_gemm_tn_multi_workspace_bytes(M, n, n1, n2)
            ws = self._workspace(nbytes)
            rc = L.slk_gemm_tn_multi_bf16x6_f32(n, A, lda, Bm, ldb, C, ldc, M, n1, n2, cs, ws.data_ptr(), nbytes, layers._stream())
            if rc != _lib.SLK_ERR_UNSUPPORTED:
                _lib.check(rc, "gemm_tn (several)")
                return
        for A, lda, Bm, ldb, C, ldc, n1, n2, cs in problems:
            self._tn(A, lda, Bm, ldb, C, ldc, M, n1, n2, colsum=cs)

    def _gemm(self, x, ldx, W, bias, y, ldy, M, K, N, act):
        """y = act(x . W^T + b), W:[N][K] -- fp16 3-term split where it applies, else float32 MFMA."""
        import torch
        L = _lib.lib()
        rc = _lib.SLK_ERR_UNSUPPORTED
        if layers.SPLIT_F16 and K <= 192 and N <= 2048:
            kp = (K + 15) // 16 * 16
            hi = torch.empty((N, kp), dtype=torch.float16, device=W.device)
            lo = torch.empty((N, kp), dtype=torch.float16, device=W.device)
            inv = torch.empty((N,), dtype=torch.float32, device=W.device)
            _lib.check(L.slk_split_f16x2_f32(W.data_ptr(), N, K, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), layers._stream()),
                       "split")
            rc = L.slk_gemm_bias_act_f16x3(x, ldx, hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), bias, y, ldy, M, K, N, act,
                                           layers._stream())
        if rc == _lib.SLK_ERR_UNSUPPORTED:
            rc = L.slk_gemm_bias_act_f32(x, ldx, W.data_ptr(), bias, y, ldy, M, K, N, act, layers._stream())
        _lib.check(rc, "gemm")

    # ---- the step ---------------------------------------------------------------------------------------------------
    def __call__(self, x, labels, weights, rate):
        """One training step.  (Loss and accuracy are read back BEFORE the update is queued, although that leaves the device idle for
        60-85 us: a batch whose labels are out of range must raise without having changed a parameter.)"""
        loss, acc = self.forward_backward(x, labels, weights)
        self.update(rate)
        return loss, acc

    def forward_backward(self, x, labels, weights):
        """Loss and accuracy of the (global) batch; leaves d loss / d params (without the l2 term, which the update adds) in
        self.grad, summed over the ranks, and the factor that turns the sum into the mean in self.gscale."""
        self._queue_forward_backward(x, labels, weights)
        return self._read_loss()

    def _read_loss(self):
        """(loss, accuracy) of the step that _queue_forward_backward queued; raises if its labels were out of range (checked on the
        device beside the step -- a host-side check was two more synchronisations in the middle of it)."""
        s = self._scalars.cpu().numpy()
        if s[3] != 0.0:                                  # (summed over the ranks: every rank raises, none updates)
            raise ValueError("labels must lie in [0, %d)" % self.softmax.size)
        loss = float(s[0]) * self.gscale + (self.l2 * float(s[2]) if self.l2 != 0.0 else 0.0)
        return loss, float(s[1]) * self.gscale

    def _queue_forward_backward(self, x, labels, weights):
        import torch
        from . import device as D
        L = _lib.lib()
        st = layers._stream
        x = D.to_dev(x)
        if x.dim() != 3 or x.shape[2] != self.network.insize:
            raise ValueError("x must be [T, B, %d]" % self.network.insize)
        T, B = int(x.shape[0]), int(x.shape[1])
        # ---- forward: the inference kernels, keeping every layer's output -------------------------------------------
        h_top, tapes = x, []
        for sub in self.body:
            h_top, tp = self._forward(sub, h_top, False)
            tapes.append(tp)
        sm = self.softmax
        To = int(h_top.shape[0])
        M = To * B
        labels = D.to_dev(np.ascontiguousarray(labels) if not isinstance(labels, torch.Tensor) else labels, torch.int32)
        weights = D.to_dev(weights)
        if tuple(labels.shape) != (To, B) or tuple(weights.shape) != (To, B):
            raise ValueError("labels and weights must be [%d, %d] (the network's output length x batch)" % (To, B))
        if 2 * self.drop >= To:
            raise ValueError("drop=%d leaves nothing of %d output steps" % (self.drop, To))
        # ---- loss, accuracy, d loss / d logits (in place) -------------------------------------------------------------
        rows = torch.empty((2, M), dtype=torch.float32, device=x.device)
        sc = self._scalars
        # labels outside [0, size): the two-pass kernel only compares columns with the label (a row without a match gets garbage, no
        # access outside its buffers), so the check runs on the device and is read with the loss; the in-place kernel indexes the
        # logits row with the label and needs the answer first
        bad = ((labels < 0) | (labels >= sm.size)).any()
        sc[3] = bad                                      # travels with the sums: allreduce_step_scalars_ below, _read_loss
        logits, ld = self._softmax_grad_two_pass(h_top, labels, weights, rows)
        if logits is None:
            # (with several ranks the others learn of it through the flag: this rank must not leave the step's collectives to them)
            if rank_and_world()[1] == 1 and bool(bad.item()):
                raise ValueError("labels must lie in [0, %d)" % sm.size)
            if rank_and_world()[1] > 1:
                labels = labels.clamp(0, sm.size - 1)    # the in-place kernel indexes the logits row with the label
            logits, stats, ld = sm.logits_and_stats(h_top)
            with profiler.region("train_xent", 0.0, 8.0 * M * ld):
                _lib.check(L.slk_softmax_xent_grad_f32(logits.data_ptr(), ld, stats.data_ptr(), labels.data_ptr(),
                                                       weights.data_ptr(), To, B, sm.size, self.drop, self.min_prob,
                                                       rows[0].data_ptr(), rows[1].data_ptr(), st()), "softmax_xent")
        with profiler.region("train_xent_sums", 0.0, 8.0 * M):
            _lib.check(L.slk_reduce_rows_sum_f32(rows.data_ptr(), 2, M, sc.data_ptr(), self._sum_scratch.data_ptr(), st()), "reduce")
            if self.l2 != 0.0:
                _lib.check(L.slk_reduce_sum_f32(self.flat.data_ptr(), self.flat.numel(), 1, sc[2:].data_ptr(), st()), "reduce")
        # ---- softmax layer ---------------------------------------------------------------------------------------------
        n_in = sm.insize
        with profiler.region("train_wgrad", 2.0 * M * sm.size * n_in, 4.0 * M * (ld + n_in)):
            self._tn(logits.data_ptr(), ld, h_top.data_ptr(), layers._row_stride(h_top), self._grad_of(sm.W).data_ptr(), n_in,
                     M, sm.size, n_in, colsum=self._grad_of(sm.b).data_ptr() if sm.has_bias else None)
        dy = None
        if self.body:
            wt = torch.zeros((n_in, ld), dtype=torch.float32, device=x.device)          # W^T, rows padded like the logits
            wt[:, :sm.size] = sm.W.dev().t()
            dy = torch.empty((To, B, n_in), dtype=torch.float32, device=x.device)
            with profiler.region("train_dx", 2.0 * M * sm.size * n_in, 4.0 * M * (ld + n_in)):
                _dx_gemm(logits.data_ptr(), ld, wt, dy.data_ptr(), n_in, M, "softmax dx")
        del logits
        # ---- the layers in front of it, top down ----------------------------------------------------------------------
        for k in range(len(tapes) - 1, -1, -1):
            # a Gru layer directly above a Convolution / FeedForward layer: its dL/dx product takes the derivative of that layer's
            # activation along (csrc/gemm_bf16x6.hip, slk_gemm_dact_bf16x6) and hands down dL/d(pre-activation)
            self._below, self._dpre_ready = None, False
            if k > 0 and tapes[k][0] == "gru" and tapes[k - 1][0] in ("conv", "ff"):
                lb = tapes[k - 1][1]
                try:
                    if activation.act_name(lb.fun) in _FF_ACTS:
                        self._below = (tapes[k - 1][4], activation.act_id(lb.fun))
                except (KeyError, ValueError, AttributeError):
                    pass
            dy = self._backward(tapes[k], dy, need_dx=k > 0)
            if self._dpre_ready and k > 0:
                tapes[k - 1] = tapes[k - 1] + ("dpre",)             # the layer below skips its slk_act_backward_f32
            self._below = None
            tapes[k] = None
        # ---- data-parallel average --------------------------------------------------------------------------------------
        self.gscale = allreduce_mean_(self.grad)
        allreduce_step_scalars_(sc)

    def _softmax_grad_two_pass(self, h_top, labels, weights, rows):
        """(d loss / d logits [M][ld], ld) with the rows' loss and accuracy terms in rows[0], rows[1] -- or (None, None) where the
        two-pass kernel does not apply (csrc/gemm_rows_f16x3.hip: the layer's products are computed twice and the logits never
        exist in memory; 4 M ld bytes of traffic where logits -> gradient in place -> two readers move 12 M ld)."""
        import torch
        sm = self.softmax
        if not (sm.split_f16 and sm.insize <= 128 and sm.size <= 2048) or not XENT_TWO_PASS:
            return None, None
        To, B = int(h_top.shape[0]), int(h_top.shape[1])
        M, L = To * B, _lib.lib()
        ld = ((sm.size + 31) // 32) * 32
        grad = torch.empty((M, ld), dtype=torch.float32, device=h_top.device)
        xrow = torch.empty((M, 4), dtype=torch.float32, device=h_top.device)
        hi, lo, inv = sm._split_weights()
        flops = 2.0 * M * sm.insize * sm.size
        with profiler.region("train_softmax_xent", 2.0 * flops, 4.0 * M * (2 * sm.insize + ld), f16x3_flops=2.0 * flops):
            rc = L.slk_linear_xent_grad_f16x3(h_top.data_ptr(), layers._row_stride(h_top), hi.data_ptr(), lo.data_ptr(), inv.data_ptr(),
                                              sm.b.dev().data_ptr(), grad.data_ptr(), ld, sm.insize, sm.size, labels.data_ptr(),
                                              weights.data_ptr(), To, B, self.drop, self.min_prob, rows[0].data_ptr(),
                                              rows[1].data_ptr(), xrow.data_ptr(), layers._stream())
        if rc == _lib.SLK_ERR_UNSUPPORTED:
            return None, None
        _lib.check(rc, "softmax_xent (two passes)")
        return grad, ld

    def _forward(self, layer, x, rev):
        """(output, tape): run `layer` on the inference kernels and keep what its reverse pass needs."""
        import torch
        layer, rev = _unwrap(layer, rev)
        if isinstance(layer, layers.Serial):
            tapes = []
            for sub in layer.layers:
                x, tp = self._forward(sub, x, rev)
                tapes.append(tp)
            return x, ("serial", tapes)
        if isinstance(layer, layers.Parallel):
            subs = [_unwrap(sub, rev) for sub in layer.layers]
            if all(isinstance(sub, (layers.Gru, layers.Lstm, layers.FeedForward)) for sub, _ in subs):
                # the sub-layers write their slices of the concatenated output directly (and run side by side on their
                # own streams at small batches): their outputs are strided views of it
                y = layer._forward(x, None, rev)
                tapes, off = [], 0
                for sub, srev in subs:
                    kind = "gru" if isinstance(sub, layers.Gru) else "lstm" if isinstance(sub, layers.Lstm) else "ff"
                    tapes.append((kind, sub, srev, x, y[:, :, off:off + sub.size]))
                    off += sub.size
                return y, ("parallel", tapes, [sub.size for sub, _ in subs])
            outs, tapes = [], []
            for sub in layer.layers:
                ysub, tp = self._forward(sub, x, rev)
                outs.append(ysub)
                tapes.append(tp)
            return torch.cat(outs, dim=2), ("parallel", tapes, [int(o.shape[2]) for o in outs])
        kind = ("conv" if isinstance(layer, layers.Convolution) else "gru" if isinstance(layer, layers.Gru) else
                "lstm" if isinstance(layer, layers.Lstm) else "window" if isinstance(layer, layers.Window) else "ff")
        x = layers._check_input(x, layer.insize)
        if kind == "gru":
            saved = self._gru_forward_saving(layer, x, rev)
            if saved is not None:
                return saved[0], (kind, layer, rev, x, saved[0], saved[1])
        y = layer._forward(x, None, rev)
        return y, (kind, layer, rev, x, y)

    def _backward(self, tape, dy, need_dx):
        """dL/d(input) of the sub-tree `tape` describes (None when not needed); parameter gradients go to self.grad."""
        kind = tape[0]
        if kind == "serial":
            tapes = tape[1]
            for k in range(len(tapes) - 1, -1, -1):
                dy = self._backward(tapes[k], dy, need_dx or k > 0)
            return dy
        if kind == "parallel":
            _, tapes, sizes = tape
            dx, off = None, 0
            for tp, size in zip(tapes, sizes):
                d = self._backward(tp, dy[:, :, off:off + size], need_dx)
                off += size
                if need_dx:
                    if dx is None:
                        dx = d
                    else:
                        _lib.check(_lib.lib().slk_add_inplace_f32(dx.data_ptr(), d.data_ptr(), dx.numel(), layers._stream()),
                                   "add")
            return dx
        layer, rev, xin, y = tape[1:5]
        self._dy_is_dpre = isinstance(tape[-1], str) and tape[-1] == "dpre"
        if kind == "gru":
            return self._gru_backward(layer, rev, xin, y, dy, need_dx, saved=tape[5] if len(tape) > 5 else None)
        if kind == "lstm":
            return self._lstm_backward(layer, rev, xin, y, dy, need_dx)
        if kind == "ff":
            return self._ff_backward(layer, xin, y, dy, need_dx)
        if kind == "window":                            # first layer, no parameters: nothing to do
            assert not need_dx
            return None
        return self._conv_backward(layer, xin, y, dy, need_dx)

    def _ff_backward(self, layer, xin, y, dy, need_dx):
        """FeedForward (layers.py:157-158): dpre = dy * fun'(.), dW = dpre^T x, db = dpre^T 1, dx = dpre . W"""
        import torch
        L = _lib.lib()
        st = layers._stream
        T, B, n = int(dy.shape[0]), int(dy.shape[1]), layer.size
        M, i_sz = T * B, layer.insize
        y, dy = y.contiguous(), dy.contiguous()
        if self._dy_is_dpre:                             # the layer above has applied fun'(.) already (_dx_gemm, below=...)
            dpre = dy.reshape(M, n)
        else:
            dpre = torch.empty((M, n), dtype=torch.float32, device=dy.device)
            _lib.check(L.slk_act_backward_f32(dy.data_ptr(), y.data_ptr(), dpre.data_ptr(), dpre.numel(),
                                              activation.act_id(layer.fun), st()), "act_backward")
        with profiler.region("train_wgrad", 2.0 * M * n * i_sz, 4.0 * M * (n + i_sz)):
            self._tn(dpre.data_ptr(), n, xin.data_ptr(), layers._row_stride(xin), self._grad_of(layer.W).data_ptr(), i_sz, M, n,
                     i_sz, colsum=self._grad_of(layer.b).data_ptr() if layer.has_bias else None)
        if not need_dx:
            return None
        dx = torch.empty((T, B, i_sz), dtype=torch.float32, device=dy.device)
        with profiler.region("train_dx", 2.0 * M * n * i_sz, 4.0 * M * (n + i_sz)):
            _dx_gemm(dpre.data_ptr(), n, layer.W.dev().t().contiguous(), dx.data_ptr(), i_sz, M, "ff dx")
        return dx

    def _gru_forward_saving(self, layer, x, rev):
        """Forward pass of a Gru layer through the fused kernel's training instantiation, which also writes the gates
        [z | r] of every step.  The output lives inside a buffer with one zero time step on either side, so "h at the
        previous scan step" is just a shifted view of it (no packing pass).  None when no such instantiation exists
        (sizes, activations, SLOIKA_AMD_EXACT_F32): the caller then runs the plain forward pass and the reverse pass
        recomputes the gates."""
        import torch
        if not layers.SPLIT_F16:
            return None
        T, B, n = int(x.shape[0]), int(x.shape[1]), layer.size
        hbuf = torch.empty((T + 2, B, n), dtype=torch.float32, device=x.device)
        hbuf[0].zero_()
        hbuf[T + 1].zero_()
        y = hbuf[1:T + 1]
        zr = torch.empty((T * B, 2 * n), dtype=torch.float32, device=x.device)
        M = T * B
        rc = _lib.SLK_ERR_UNSUPPORTED
        if layers.RECURRENT_F16:            # projection and recurrence as fp16 splits (csrc/gru_bar16.hip)
            # (as layers.Gru.run prices it: the projection three fp16 MFMAs per product, the recurrent products two up to eight chunks
            #  per workgroup -- training batches of up to 2048 chunks on 256 CUs -- and three on the sixteen-chunk plan)
            two_term_fw = (B + 7) // 8 <= layers._cu_count(x.device)
            with profiler.region("gru_fused", 6.0 * M * n * (n + layer.insize), 4.0 * M * (layer.insize + 3 * n),
                                 f16x3_flops=6.0 * M * n * (layer.insize if two_term_fw else n + layer.insize),
                                 f16x2_flops=6.0 * M * n * n if two_term_fw else 0.0) as reg:
                rc = layers.gru_f16_entry()(x.data_ptr(), layers._row_stride(x), layer.iW.dev().data_ptr(),
                                            layer.sW.dev().data_ptr(), layer.sW2.dev().data_ptr(),
                                            layer.b.dev().data_ptr(), y.data_ptr(), n, T, B, layer.insize, n, int(rev),
                                            activation.act_id(layer.fun), activation.act_id(layer.gatefun), None,
                                            zr.data_ptr(), layers._stream())
                if rc == _lib.SLK_ERR_UNSUPPORTED and reg is not None:
                    reg.cancel()
        if rc == _lib.SLK_ERR_UNSUPPORTED:
            return None
        _lib.check(rc, "gru_fused_train")
        hprev = hbuf[2:T + 2] if rev else hbuf[0:T]
        return y, (zr, hprev)

    def _gru_backward(self, layer, rev, xin, h, dy, need_dx, saved=None):
        """Reverse pass of one Gru layer.  Sizes that are not multiples of 16 (models/raw_1.00_rGr.py: 110, 142) run
        zero-padded, like the forward pass (layers.Gru._padded): padding neurons have zero weights, zero state and receive
        zero gradient, so the leading blocks of every gradient are those of the unpadded layer."""
        import torch
        n, i_sz = layer.size, layer.insize
        n16, i16 = (n + 15) // 16 * 16, (i_sz + 15) // 16 * 16
        iW, sW, sW2, b = layer.iW.dev(), layer.sW.dev(), layer.sW2.dev(), layer.b.dev()
        gb = self._grad_of(layer.b) if layer.has_bias else None
        if n16 == n and i16 == i_sz:
            return self._gru_backward_core(xin, h, dy, rev, n, i_sz, iW, sW, sW2, b, self._grad_of(layer.iW),
                                           self._grad_of(layer.sW), self._grad_of(layer.sW2), gb, need_dx, saved=saved)
        T, B, dev = int(h.shape[0]), int(h.shape[1]), h.device

        def widen(t, width):
            out = torch.zeros((T, B, width), dtype=torch.float32, device=dev)
            out[:, :, :t.shape[2]] = t
            return out

        def pad(t, blocks, rows, cols, prow, pcol):
            out = torch.zeros((blocks, prow, pcol), dtype=torch.float32, device=dev)
            out[:, :rows, :cols] = t.reshape(blocks, rows, cols)
            return out.reshape(blocks * prow, pcol)

        iW_p, sW_p, sW2_p = pad(iW, 3, n, i_sz, n16, i16), pad(sW, 2, n, n, n16, n16), pad(sW2, 1, n, n, n16, n16)
        b_p = pad(b, 3, n, 1, n16, 1).reshape(-1)
        giW, gsW, gsW2 = torch.empty_like(iW_p), torch.empty_like(sW_p), torch.empty_like(sW2_p)
        gb_p = torch.empty_like(b_p) if gb is not None else None
        dx = self._gru_backward_core(widen(xin, i16), widen(h, n16), widen(dy, n16), rev, n16, i16, iW_p, sW_p, sW2_p, b_p,
                                     giW.reshape(-1), gsW.reshape(-1), gsW2.reshape(-1), gb_p, need_dx)
        self._grad_of(layer.iW).view(3, n, i_sz).copy_(giW.view(3, n16, i16)[:, :n, :i_sz])
        self._grad_of(layer.sW).view(2, n, n).copy_(gsW.view(2, n16, n16)[:, :n, :n])
        self._grad_of(layer.sW2).view(n, n).copy_(gsW2.view(n16, n16)[:n, :n])
        if gb is not None:
            gb.view(3, n).copy_(gb_p.view(3, n16)[:, :n])
        return dx[:, :, :i_sz].contiguous() if need_dx else None

    def _gru_backward_core(self, xin, h, dy, rev, n, i_sz, iW, sW, sW2, b, giW, gsW, gsW2, gb, need_dx, saved=None):
        import torch
        L = _lib.lib()
        st = layers._stream
        T, B = int(h.shape[0]), int(h.shape[1])
        M, K = T * B, i_sz + n
        act, gact = activation.act_id(activation.tanh), activation.act_id(activation.sigmoid)
        dev = h.device
        if saved is not None:
            # the forward pass left the gates and a shifted view of its output (see _gru_forward_saving)
            zr, hprev = saved
            x_ptr, ldx, hp_ptr, ldhp = xin.data_ptr(), layers._row_stride(xin), hprev.data_ptr(), layers._row_stride(hprev)
        else:
            # recompute [z r] of all steps at once: one GEMM over packed rows [x_t | h_{t-1}]
            with profiler.region("train_gates", 4.0 * M * n * K, 4.0 * M * (2 * K + 2 * n), f16x3_flops=4.0 * M * n * K):
                xh = torch.empty((M, K), dtype=torch.float32, device=dev)
                _lib.check(L.slk_train_pack_xh_f32(xin.data_ptr(), layers._row_stride(xin), h.data_ptr(),
                                                   layers._row_stride(h), xh.data_ptr(), T, B, i_sz, n, int(rev), st()),
                           "pack_xh")
                zr = torch.empty((M, 2 * n), dtype=torch.float32, device=dev)
                self._gemm(xh.data_ptr(), K, torch.cat([iW[:2 * n], sW], 1).contiguous(), b[:2 * n].data_ptr(), zr.data_ptr(),
                           2 * n, M, K, 2 * n, gact)
            x_ptr, ldx, hp_ptr, ldhp = xh.data_ptr(), K, xh.data_ptr() + 4 * i_sz, K
        # the candidate is not recomputed by a GEMM: the scan recovers it from the layer's own output (csrc/train.hip)
        da = torch.empty((M, 3 * n), dtype=torch.float32, device=dev)
        rh = torch.empty((M, n), dtype=torch.float32, device=dev)
        # gru_bwd16_kernel issues TWO fp16 MFMAs per product (v_mfma_f32_16x16x32_f16, hi and lo halves of the operand in different
        # column groups): the region says so, and bench.py prices it on the fp16 pipe at two instructions per product
        two_term = layers.SPLIT_F16 and layers.RECURRENT_F16
        # dL/dx out of the scan itself (csrc/gru_bwd16.hip, DX: the operand images of a step are da of that step, so the product costs the
        # pass 18 MFMAs per wave and step and no second reading of da) -- unless the layer below hands its activation's derivative to the
        # product's epilogue (slk_gemm_dact_bf16x6), which the separate GEMM keeps
        dx = None
        if need_dx and two_term and "no_scan_dx" not in layers._DEBUG:
            # the layer below's activation (a Convolution / FeedForward layer whose OUTPUT is the very tensor this Gru consumed: same
            # memory, rows a uniform distance apart -- the condition of the separate product's fused form further down)
            yb_ptr, ldyb, dact = None, 0, 0
            if self._below is not None:
                yb = self._below[0]
                if (tuple(yb.shape) == (T, B, i_sz) and yb.stride(2) == 1 and yb.stride(0) == B * yb.stride(1)
                        and yb.data_ptr() == xin.data_ptr() and tuple(yb.stride()) == tuple(xin.stride())):
                    yb_ptr, ldyb, dact = yb.data_ptr(), layers._row_stride(yb), self._below[1]
            dx = torch.empty((T, B, i_sz), dtype=torch.float32, device=dev)
            with profiler.region("train_gru_scan", 6.0 * M * n * (n + i_sz), 4.0 * M * (9 * n + i_sz),
                                 f16x2_flops=6.0 * M * n * (n + i_sz)) as reg_dx:
                rc = _lib.SLK_ERR_UNSUPPORTED              # (a layer below whose output is not this Gru's input tensor: the separate pair)
                if self._below is None or yb_ptr is not None:
                    rc = L.slk_gru_backward16_dx_f32(dy.data_ptr(), layers._row_stride(dy), hp_ptr, ldhp, zr.data_ptr(), h.data_ptr(),
                                                     layers._row_stride(h), sW.data_ptr(), sW2.data_ptr(), iW.data_ptr(), da.data_ptr(),
                                                     rh.data_ptr(), dx.data_ptr(), i_sz, T, B, n, i_sz, int(rev), act, gact, yb_ptr, ldyb,
                                                     dact, st())
                if rc == _lib.SLK_ERR_UNSUPPORTED and reg_dx is not None:
                    reg_dx.cancel()
            if rc == _lib.SLK_ERR_UNSUPPORTED:
                dx = None
            else:
                _lib.check(rc, "gru_backward (with dx)")
                if yb_ptr is not None:
                    self._below, self._dpre_ready = None, True         # what leaves IS dL/d(pre-activation) of the layer below
        with profiler.region("train_gru_scan", 6.0 * M * n * n, 4.0 * M * 9 * n, f16x2_flops=6.0 * M * n * n if two_term else 0.0) as reg_scan:
            rc = _lib.SLK_OK if dx is not None else _lib.SLK_ERR_UNSUPPORTED
            if dx is not None and reg_scan is not None:
                reg_scan.cancel()
            if dx is None and layers.SPLIT_F16 and layers.RECURRENT_F16:     # the two products of a step as fp16 splits (csrc/gru_bwd16.hip: n <= 128)
                rc = L.slk_gru_backward16_f32(dy.data_ptr(), layers._row_stride(dy), hp_ptr, ldhp, zr.data_ptr(),
                                              h.data_ptr(), layers._row_stride(h), sW.data_ptr(), sW2.data_ptr(), da.data_ptr(),
                                              rh.data_ptr(), T, B, n, int(rev), act, gact, st())
            if rc == _lib.SLK_ERR_UNSUPPORTED:
                rc = L.slk_gru_backward_f32(dy.data_ptr(), layers._row_stride(dy), hp_ptr, ldhp, zr.data_ptr(),
                                            h.data_ptr(), layers._row_stride(h), sW.data_ptr(), sW2.data_ptr(), da.data_ptr(),
                                            rh.data_ptr(), T, B, n, int(rev), act, gact, st())
        if rc == _lib.SLK_ERR_UNSUPPORTED:
            raise NotImplementedError("training: no reverse-scan kernel for a Gru of size %d" % n)
        _lib.check(rc, "gru_backward")
        f4 = 4                                                                       # bytes per float, for column offsets
        with profiler.region("train_wgrad", 2.0 * M * (3 * n * i_sz + 3 * n * n), 4.0 * M * (3 * n + 2 * K)):
            self._tn_many([(da.data_ptr(), 3 * n, x_ptr, ldx, giW.data_ptr(), i_sz, 3 * n, i_sz, gb.data_ptr() if gb is not None else None),
                           (da.data_ptr(), 3 * n, hp_ptr, ldhp, gsW.data_ptr(), n, 2 * n, n, None),
                           (da.data_ptr() + f4 * 2 * n, 3 * n, rh.data_ptr(), n, gsW2.data_ptr(), n, n, n, None)], M)
        if not need_dx:
            return None
        if dx is not None:
            return dx
        dx = torch.empty((T, B, i_sz), dtype=torch.float32, device=dev)
        with profiler.region("train_dx", 6.0 * M * n * i_sz, 4.0 * M * (3 * n + i_sz)):
            below, self._below = self._below, None
            # the fused product reads the layer below's OUTPUT row by row with one row stride: only if it is the very tensor this Gru
            # consumed (same memory, rows a uniform distance apart); a copied or converted input, or rows of uneven pitch, take the
            # separate slk_act_backward_f32 pass instead
            if below is not None:
                yb = below[0]
                if (tuple(yb.shape) != (T, B, i_sz) or yb.stride(2) != 1 or yb.stride(0) != B * yb.stride(1)
                        or yb.data_ptr() != xin.data_ptr() or tuple(yb.stride()) != tuple(xin.stride())):
                    below = None
            self._dpre_ready = _dx_gemm(da.data_ptr(), 3 * n, iW.t().contiguous(), dx.data_ptr(), i_sz, M, "gru dx", below=below)
        return dx

    #: Lstm widths the reverse-scan kernels are instantiated for (csrc/train.hip); others run zero-padded to the next one
    _LSTM_SCAN_SIZES = (16, 32, 48, 64, 96, 128)

    def _lstm_backward(self, layer, rev, xin, out, dy, need_dx):
        """Reverse pass of one Lstm layer.  Widths without a reverse-scan instantiation run zero-padded to the next one that
        has: a padding neuron has zero weights, bias and peepholes, so its gates are gatefun(0), its cell and output stay 0
        and it receives and passes on zero gradient -- the leading blocks of every gradient are those of the unpadded layer
        (rows of iW / sW / b are interleaved neuron-major, j*4 + gate, so padding appends rows)."""
        import torch
        n, i_sz = layer.size, layer.insize
        iW, sW, b = layer.iW.dev(), layer.sW.dev(), layer.b.dev()
        peep = layer.p.dev() if layer.has_peep else None
        gb = self._grad_of(layer.b) if layer.has_bias else None
        gp = self._grad_of(layer.p) if layer.has_peep else None
        act, gact = activation.act_id(layer.fun), activation.act_id(layer.gatefun)
        if n in self._LSTM_SCAN_SIZES:
            return self._lstm_backward_core(xin, out, dy, rev, n, i_sz, iW, sW, b, peep, self._grad_of(layer.iW),
                                            self._grad_of(layer.sW), gb, gp, need_dx, act, gact)
        bigger = [m for m in self._LSTM_SCAN_SIZES if m >= n]
        if not bigger:
            raise NotImplementedError("training: no reverse-scan kernel for an Lstm of size %d" % n)
        npad = bigger[0]
        T, B, dev = int(out.shape[0]), int(out.shape[1]), out.device

        def widen(t):
            w = torch.zeros((T, B, npad), dtype=torch.float32, device=dev)
            w[:, :, :n] = t
            return w

        iW_p = torch.zeros((4 * npad, i_sz), dtype=torch.float32, device=dev)
        iW_p[:4 * n] = iW.reshape(4 * n, i_sz)
        sW_p = torch.zeros((4 * npad, npad), dtype=torch.float32, device=dev)
        sW_p[:4 * n, :n] = sW.reshape(4 * n, n)
        b_p = torch.zeros(4 * npad, dtype=torch.float32, device=dev)
        b_p[:4 * n] = b.reshape(-1)
        peep_p = None
        if peep is not None:
            peep_p = torch.zeros((3, npad), dtype=torch.float32, device=dev)
            peep_p[:, :n] = peep.reshape(3, n)
        giW, gsW = torch.empty_like(iW_p), torch.empty_like(sW_p)
        gb_p = torch.empty_like(b_p) if gb is not None else None
        gp_p = torch.empty((3, npad), dtype=torch.float32, device=dev) if gp is not None else None
        dx = self._lstm_backward_core(xin, widen(out), widen(dy), rev, npad, i_sz, iW_p, sW_p, b_p, peep_p, giW.reshape(-1),
                                      gsW.reshape(-1), gb_p, None if gp_p is None else gp_p.reshape(-1), need_dx, act, gact)
        self._grad_of(layer.iW).view(4 * n, i_sz).copy_(giW[:4 * n])
        self._grad_of(layer.sW).view(4 * n, n).copy_(gsW[:4 * n, :n])
        if gb is not None:
            gb.view(-1).copy_(gb_p[:4 * n])
        if gp is not None:
            gp.view(3, n).copy_(gp_p[:, :n])
        return dx

    def _lstm_backward_core(self, xin, out, dy, rev, n, i_sz, iW, sW, b, peep_t, giW, gsW, gb, gp, need_dx, act, gact):
        """layers.py:677-697 differentiated: gate inputs of all steps as one GEMM over [x_t | out_{t-1}], the element-wise cell
        recursion, the reverse scan, then the weight gradients as contractions over all rows."""
        import torch
        L = _lib.lib()
        st = layers._stream
        T, B = int(out.shape[0]), int(out.shape[1])
        M, K = T * B, i_sz + n
        dev = out.device
        peep = peep_t.data_ptr() if peep_t is not None else None
        with profiler.region("train_gates", 8.0 * M * n * K, 4.0 * M * (K + 8 * n)):
            xh = torch.empty((M, K), dtype=torch.float32, device=dev)
            _lib.check(L.slk_train_pack_xh_f32(xin.data_ptr(), layers._row_stride(xin), out.data_ptr(), layers._row_stride(out),
                                               xh.data_ptr(), T, B, i_sz, n, int(rev), st()), "pack_xh")
            summed = torch.empty((M, 4 * n), dtype=torch.float32, device=dev)
            self._gemm(xh.data_ptr(), K, torch.cat([iW.reshape(4 * n, i_sz), sW.reshape(4 * n, n)], 1).contiguous(), b.data_ptr(),
                       summed.data_ptr(), 4 * n, M, K, 4 * n, 0)
            gates = torch.empty((M, 4 * n), dtype=torch.float32, device=dev)
            cell = torch.empty((M, n), dtype=torch.float32, device=dev)
            _lib.check(L.slk_lstm_gates_f32(summed.data_ptr(), peep, gates.data_ptr(), cell.data_ptr(), T, B, n, int(rev),
                                            st()), "lstm_gates")
        dsum = summed                                                            # reuse: the sums are not needed again
        dpeep = torch.empty((B, 3 * n), dtype=torch.float32, device=dev)
        with profiler.region("train_lstm_scan", 8.0 * M * n * n, 4.0 * M * 10 * n):
            rc = _lib.SLK_ERR_UNSUPPORTED
            if layers.SPLIT_F16 and layers.RECURRENT_F16:     # the product of a step as an fp16 split (csrc/lstm_bwd16.hip: n <= 64)
                rc = L.slk_lstm_backward16_f32(dy.data_ptr(), layers._row_stride(dy), gates.data_ptr(), cell.data_ptr(),
                                               sW.data_ptr(), peep, dsum.data_ptr(), dpeep.data_ptr(), T, B, n, int(rev), act, gact,
                                               st())
            if rc == _lib.SLK_ERR_UNSUPPORTED:
                rc = L.slk_lstm_backward_f32(dy.data_ptr(), layers._row_stride(dy), gates.data_ptr(), cell.data_ptr(),
                                             sW.data_ptr(), peep, dsum.data_ptr(), dpeep.data_ptr(), T, B, n, int(rev), act, gact,
                                             st())
        if rc == _lib.SLK_ERR_UNSUPPORTED:
            raise NotImplementedError("training: no reverse-scan kernel for an Lstm of size %d" % n)
        _lib.check(rc, "lstm_backward")
        with profiler.region("train_wgrad", 8.0 * M * n * K, 4.0 * M * (4 * n + K)):
            self._tn_many([(dsum.data_ptr(), 4 * n, xh.data_ptr(), K, giW.data_ptr(), i_sz, 4 * n, i_sz, gb.data_ptr() if gb is not None else None),
                           (dsum.data_ptr(), 4 * n, xh.data_ptr() + 4 * i_sz, K, gsW.data_ptr(), n, 4 * n, n, None)], M)
            if gp is not None:                        # sum of the per-chunk peephole gradients: the column sums of dpeep
                scratch = torch.empty(3 * n, dtype=torch.float32, device=dev)
                self._tn(dpeep.data_ptr(), 3 * n, dpeep.data_ptr(), 3 * n, scratch.data_ptr(), 1, B, 3 * n, 1,
                         colsum=gp.data_ptr())
        if not need_dx:
            return None
        dx = torch.empty((T, B, i_sz), dtype=torch.float32, device=dev)
        with profiler.region("train_dx", 8.0 * M * n * i_sz, 4.0 * M * (4 * n + i_sz)):
            _dx_gemm(dsum.data_ptr(), 4 * n, iW.reshape(4 * n, i_sz).t().contiguous(), dx.data_ptr(), i_sz, M, "lstm dx")
        return dx

    def _conv_backward(self, layer, xin, y, dy, need_dx=False):
        """Convolution (layers.py:417-419, conv.py:66-111) as window rows times the flattened filter bank: dpre = dy * fun'(.),
        dW = dpre^T cols, db = dpre^T 1, and for a convolution that is not the first layer dcols = dpre . W folded back onto
        the input (col2im)."""
        import torch
        L = _lib.lib()
        st = layers._stream
        T, B = int(xin.shape[0]), int(xin.shape[1])
        To, n, cin, w = int(y.shape[0]), layer.size, layer.insize, layer.winlen
        M, K = To * B, layer.insize * layer.winlen
        y, dy = y.contiguous(), dy.contiguous()
        if self._dy_is_dpre:                             # the layer above has applied fun'(.) already (_dx_gemm, below=...)
            dpre = dy
        else:
            dpre = torch.empty_like(y)
            rc = L.slk_act_backward_f32(dy.data_ptr(), y.data_ptr(), dpre.data_ptr(), y.numel(), activation.act_id(layer.fun), st())
            if rc == _lib.SLK_ERR_UNSUPPORTED:
                raise NotImplementedError("training: Convolution activation %s has no derivative kernel" % layer.fun.__name__)
            _lib.check(rc, "act_backward")
        cols = torch.empty((M, K), dtype=torch.float32, device=y.device)
        xc = xin.contiguous()
        if cin == 1:
            _lib.check(L.slk_train_im2col_cin1_f32(xc.data_ptr(), B, 1, T, B, w, layer.stride, layer.padding[0], layer.padding[1],
                                                   cols.data_ptr(), st()), "im2col")
        else:
            _lib.check(L.slk_train_im2col_f32(xc.data_ptr(), cin, T, B, cin, w, layer.stride, layer.padding[0], layer.padding[1],
                                              cols.data_ptr(), st()), "im2col")
        with profiler.region("train_wgrad", 2.0 * M * n * K, 4.0 * M * (n + K)):
            self._tn(dpre.data_ptr(), n, cols.data_ptr(), K, self._grad_of(layer.W).data_ptr(), K, M, n, K,
                     colsum=self._grad_of(layer.b).data_ptr() if layer.has_bias else None)
        if not need_dx:
            return None
        dcols = cols                                                        # reuse: the window rows are not needed again
        dx = torch.empty((T, B, cin), dtype=torch.float32, device=y.device)
        with profiler.region("train_dx", 2.0 * M * n * K, 4.0 * M * (n + K)):
            _lib.check(L.slk_gemm_bias_act_f32(dpre.data_ptr(), n, layer.W.dev().reshape(n, K).t().contiguous().data_ptr(), None,
                                               dcols.data_ptr(), K, M, n, K, 0, st()), "conv dcols")
            _lib.check(L.slk_train_col2im_f32(dcols.data_ptr(), T, B, cin, w, layer.stride, layer.padding[0], layer.padding[1],
                                              dx.data_ptr(), cin, st()), "col2im")
        return dx

    def update(self, rate):
        """One optimiser step on the gradient left by forward_backward (updates.py:36-89, or :9-33 for sgd)."""
        L = _lib.lib()
        n = self.flat.numel()
        gscale = getattr(self, "gscale", 1.0)
        if self.optimiser == "adam":
            lr_t, mdecay, self.t = adamski_scalars(self.t, rate, self.decay, self.mrate)
            rc = L.slk_adamski_update_f32(self.flat.data_ptr(), self.grad.data_ptr(), self.momentum.data_ptr(),
                                          self.variance.data_ptr(), n, lr_t, mdecay, self.decay[0], self.decay[1], self.epsilon,
                                          self.clip, self.l2, gscale, layers._stream())
        else:
            rc = L.slk_sgd_update_f32(self.flat.data_ptr(), self.grad.data_ptr(), self.momentum.data_ptr(), n, float(rate),
                                      self.sgd_momentum, self.clip, self.l2, gscale, layers._stream())
        _lib.check(rc, "update")
        self._drop_caches()

    def gradients(self):
        """d loss / d params of the last forward_backward as numpy arrays in network.params() order (updates.py:66),
        including the l2 term and the rank average -- what the optimiser sees before clipping."""
        g = (self.grad * getattr(self, "gscale", 1.0) + 2.0 * self.l2 * self.flat).cpu().numpy()
        return [g[a:b].reshape(p.shape) for p, a, b in zip(self.shared, self.offsets[:-1], self.offsets[1:])]


def wrap_network(network, min_prob=0.0, l2=0.0, drop=0, adam=(0.9, 0.999)):
    """train_network.py:124-142.  `adam` = (decay1, decay2), the reference's `args.adam.decay1/2`."""
    return TrainingStep(network, min_prob=min_prob, l2=l2, drop=drop, decay=tuple(adam))


def save_model(network, output, index=None, step=None):
    """train_network.py:145-152 (pickle of the network object, loadable by helpers.load_model)."""
    import os
    import pickle
    if step is not None:
        step.sync_host()
    model_file = 'model_final.pkl' if index is None else 'model_checkpoint_{:05d}.pkl'.format(index)
    with open(os.path.join(output, model_file), 'wb') as fh:
        pickle.dump(network, fh, protocol=pickle.HIGHEST_PROTOCOL)
    return os.path.join(output, model_file)


# ---------------------------------------------------------------------------------------------------------------------
# The loop around the step: bin/train_network.py:180-330 as callable functions (the reference has it inline in a script).
# ---------------------------------------------------------------------------------------------------------------------
class Logger(object):
    """Progress text to stdout (unless quiet) and, unbuffered, to the log file (train_network.py:153-170)."""

    def __init__(self, log_file_name, quiet=False):
        self.quiet = quiet
        self.fh = open(log_file_name, 'wb', buffering=0)

    def write(self, message):
        import sys
        if not self.quiet:
            sys.stdout.write(message)
            sys.stdout.flush()
        try:
            self.fh.write(message.encode('utf-8'))
        except IOError as err:
            print("Failed to write to log\n Message: {}\n Error: {}".format(message, repr(err)))


def load_chunk_file(path, reweight='weights'):
    """The training data of train_network.py:199-210 from a `.npz` with the datasets `chunkify` writes to HDF5
    (sloika/util.py:60-91: `chunks` [n, chunk_len, nfeature] f4, `labels` [n, label_len] i4, `bad` [n, label_len] i1,
    `weights` [n] f4) and the attributes `kmer` / `alphabet` as 0-d arrays.  h5py is not available on the GPU boxes; a
    chunk file converts with `np.savez(out, **{k: h5[k][:] for k in h5}, **dict(h5.attrs))`."""
    with np.load(path, allow_pickle=False) as z:
        data = {k: z[k] for k in z.files}
    alphabet = data["alphabet"].item() if "alphabet" in data else b"ACGT"       # variables.DEFAULT_ALPHABET, :259-264
    if isinstance(alphabet, str):
        alphabet = alphabet.encode("ascii")
    out = {"chunks": data["chunks"].astype(sloika_dtype), "labels": data["labels"].astype(np.int32), "bad": data["bad"],
           "kmer": int(data["kmer"]) if "kmer" in data else None, "alphabet": alphabet}
    if reweight is not None and reweight in data:                                   # :203-206
        out["weights"] = data[reweight].astype('float64')
    else:
        out["weights"] = np.ones(len(out["chunks"]), dtype='float64')
    return out


def prepare_training_data(data, transducer=True, bad=True, ilf=False):
    """train_network.py:207-252: normalised sampling weights, blank removal, bad positions, per-label weights.

    One deliberate difference: `all_labels[all_bad] = 0` (:240) indexes with the int8 array HDF5 returns, i.e. it
    fancy-indexes ROWS 0 and 1 of the labels instead of masking the flagged positions; here `bad` is used as the boolean
    mask the line was written for."""
    all_labels = np.array(data["labels"], dtype=np.int32)
    all_weights = np.asarray(data["weights"], dtype='float64')
    all_weights = all_weights / np.sum(all_weights)                                 # :207-208
    if not transducer:
        remove_blanks(all_labels)                                                   # :236-237
    if bad:
        all_labels[np.asarray(data["bad"]).astype(bool)] = 0                        # :239-240 (see above)
    if ilf:                                                                         # :242-249
        label_weights = np.zeros(np.max(all_labels) + 1, dtype='f4')
        for i, lbls in enumerate(all_labels):
            label_weights += all_weights[i] * np.bincount(lbls, minlength=len(label_weights))
        label_weights = np.reciprocal(label_weights)
        label_weights /= np.mean(label_weights)
    else:
        label_weights = np.ones(np.max(all_labels) + 1, dtype='f4')                 # :250-252
    return all_labels, all_weights, label_weights


def training_batches(all_chunks, all_labels, all_weights, label_weights, niteration, batch_size=100,
                     chunk_len_range=(0.5, 1.0), drop=20, rate=1e-3, lrdecay=5000.0, rank=0, world=1):
    """The sampler of train_network.py:213-230,288-306, drawing from numpy's global generator in the reference's order
    (seed it with np.random.seed like :180).  Yields (indata [chunk_len, batch, nfeature], labels [label_len, batch],
    weights, learning_rate) per iteration.

    Data parallel (`world` ranks seeded IDENTICALLY): every rank draws the same global batch -- same chunk length, same
    window, same chunk ids -- and keeps ids rank, rank + world, ...; the global batch is cut to a multiple of `world`, so all
    ranks hold the same number of positions (what TrainingStep's gradient average assumes) and no chunk is used twice."""
    data_chunk = all_chunks.shape[1]
    training_stride = int(np.ceil(float(all_chunks.shape[1]) / all_labels.shape[1]))            # :213
    min_chunk = 2 * drop + 1 if chunk_len_range[0] is None else int(np.around(chunk_len_range[0] * data_chunk))
    max_chunk = data_chunk if chunk_len_range[1] is None else int(np.around(chunk_len_range[1] * data_chunk))
    assert max_chunk >= min_chunk, "Min chunk size (got {}) must be <= chunk size (got {})".format(min_chunk, max_chunk)
    assert data_chunk >= max_chunk, "Max chunk size (got {}) must be <= data chunk size (got {})".format(max_chunk, data_chunk)
    assert data_chunk >= (2 * drop + 1), "Data chunk size (got {}) must be > 2 * drop (got {})".format(data_chunk, drop)
    assert min_chunk >= (2 * drop + 1), "Min chunk size (got {}) must be > 2 * drop (got {})".format(min_chunk, drop)
    max_batch_size = (all_weights > 0).sum()                                                    # :209
    for i in range(niteration):
        learning_rate = rate / (1.0 + i / lrdecay)                                              # :289
        chunk_len = np.random.randint(min_chunk, max_chunk + 1)                                 # :291-292
        chunk_len = chunk_len - (chunk_len % training_stride)
        this_batch = int(batch_size * float(max_chunk) / chunk_len)                             # :294
        start = np.random.randint(data_chunk - chunk_len + 1)                                   # :296-297
        start = start - (start % training_stride)
        label_lb = start // training_stride                                                     # :299-300
        label_ub = (start + chunk_len) // training_stride
        idx = np.sort(np.random.choice(len(all_chunks), size=min(this_batch, max_batch_size), replace=False,
                                       p=all_weights))                                          # :302-303
        if world > 1:
            if len(idx) < world:
                raise ValueError("a batch of %d chunks cannot be shared by %d ranks" % (len(idx), world))
            idx = idx[:len(idx) - len(idx) % world][rank::world]
        indata = np.ascontiguousarray(all_chunks[idx, start: start + chunk_len].transpose((1, 0, 2)))      # :304
        labels = np.ascontiguousarray(all_labels[idx, label_lb: label_ub].transpose())          # :305
        yield indata, labels, label_weights[labels], learning_rate                              # :306


def train_loop(network, data, output, niteration=50000, batch_size=100, chunk_len_range=(0.5, 1.0), drop=20,
               adam=(1e-3, 0.9, 0.999), lrdecay=5000.0, min_prob=1e-30, l2=0.0, save_every=5000, smooth=0.45, seed=None,
               transducer=True, bad=True, ilf=False, quiet=False):
    """train_network.py:180-330 for an already built network and loaded data (`load_chunk_file`): writes model.log,
    model_checkpoint_NNNNN.pkl every `save_every` iterations and model_final.pkl into `output`; returns the step."""
    import os
    import time
    rank, world = rank_and_world()
    if world > 1:
        # every rank must draw the same batches: one seed for all (rank 0's when none was given)
        import torch.distributed as dist
        box = [seed if seed is not None else int(np.random.SeedSequence().entropy % (2 ** 32))]
        dist.broadcast_object_list(box, src=0)
        seed = box[0]
    np.random.seed(seed)                                                            # :180
    if rank == 0 and not os.path.exists(output):
        os.mkdir(output)
    if world > 1:
        dist.barrier()
    # one writer: rank 0 owns model.log and the checkpoints; the other ranks log nowhere
    log = Logger(os.path.join(output, 'model.log'), quiet) if rank == 0 else Logger(os.devnull, True)
    all_labels, all_weights, label_weights = prepare_training_data(data, transducer, bad, ilf)
    # checked here, identically on every rank, so that no rank can fail alone inside a step and leave the others waiting
    # in the gradient all-reduce
    if all_labels.min() < 0 or all_labels.max() >= network.size:
        raise ValueError("labels must lie in [0, %d)" % network.size)
    fg = wrap_network(network, min_prob=min_prob, l2=l2, drop=drop, adam=adam[1:])
    total_ev = 0
    score_smoothed, acc_smoothed = ExponentialSmoother(smooth), ExponentialSmoother(smooth)
    log.write('* Dumping initial model\n')
    if rank == 0:
        save_model(network, output, 0, step=fg)                                     # :282
    t0 = time.time()
    log.write('* Training\n')
    batches = training_batches(data["chunks"], all_labels, all_weights, label_weights, niteration, batch_size,
                               chunk_len_range, drop, adam[0], lrdecay, rank=rank, world=world)
    for i, (indata, labels, weights, learning_rate) in enumerate(batches):
        fval, batch_acc = fg(indata, labels, weights, learning_rate)                # :308
        total_ev += np.size(labels)
        score_smoothed.update(float(fval))
        acc_smoothed.update(batch_acc)
        if (i + 1) % save_every == 0:                                               # :315-319
            if rank == 0:
                save_model(network, output, (i + 1) // save_every, step=fg)
            log.write('C')
        else:
            log.write('.')
        if (i + 1) % 50 == 0:                                                       # :321-328
            tn = time.time()
            dt = tn - t0
            log.write(' {:5d} {:5.3f}  {:5.2f}%  {:5.2f}s ({:.2f} kev/s)\n'.format(
                (i + 1) // 50, score_smoothed.value, 100.0 * acc_smoothed.value, dt, total_ev / 1000.0 / dt))
            total_ev = 0
            t0 = tn
    if rank == 0:
        save_model(network, output, step=fg)                                        # :330
    if world > 1:
        dist.barrier()
    return fg
