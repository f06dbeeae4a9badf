"""The hot path end to end, resident on one GPU:

    raw chunks [B, chunk_len] (HBM)  --normalise-->  --conv-->  --GRU/LSTM stack-->  --softmax-->
    --prepare_post + log + k-mer Viterbi + backtrace-->  (scores[B], paths[B,T'], lens[B])

This is the batched restatement of bin/basecall_network.py `raw` (sloika/basecall.py:88-121, 26-51) that
BASELINE.json's metric is quoted on.  All arithmetic happens in the HIP kernels behind include/sloika_amd.h;
torch only owns the buffers and the stream.
"""
import os

import numpy as np

from . import _lib, batch, decode, layers

#: decode straight from the Softmax layer's input (csrc/softmax_viterbi.hip: projection, softmax, prepare_post, log and the
#: Viterbi forward pass in one kernel, the logits never written) where that kernel applies; SLOIKA_AMD_FUSED_DECODE=0 keeps the
#: projection kernel + decoder pair
FUSED_DECODE = "no_fused_decode" not in layers._DEBUG


class Basecaller(object):
    def __init__(self, network, kmer_len=5, nbase=4, min_prob=1e-5, skip=0.0, normalisation='per-chunk', in_flight=1,
                 fused_decode=None, deterministic=True, borrow=False):
        """skip default 0.0 is the CLI default (bin/basecall_network.py:38).

        in_flight: how many batches the caller keeps in flight at a time, each on a HIP stream of its own (one Basecaller
        per stream).  With two or more, a Gru layer runs eight chunks per workgroup (csrc/gru_bar16d.hip) whenever that lets
        the layers of all the batches share the chip -- batch 1024, two in flight: 2 x 128 workgroups on 256 CUs -- instead of
        one workgroup per four chunks each taking the whole device in turn.

        deterministic (default): a chunk is called the same whatever the batch size and however many batches are in flight (the
        reference calls one read at a time: one read, one answer, basecall.py:88-121).  Every execution plan but one computes the same
        bits; the exception is the sixteen-chunk Gru plan (csrc/gru_bar16q.hip, three-term recurrent products: states equal to float32
        rounding, up to 2 % of chunks called differently), which calls of more than 2048 chunks and four batches in flight would take.
        With deterministic=True they run eight chunks per workgroup instead; deterministic=False lets the faster plan in (the
        price of the switch is in the bench line: `in_flight.deterministic`)."""
        self.deterministic = bool(deterministic)
        # borrow: the reference compiles its networks with In(borrow=True) / Out(borrow=True) (layers.py:34-36): what a call returns
        # may be overwritten by a later call, the caller consumes or copies it first.  Here: every buffer of a call (layer outputs,
        # workspaces, results) comes out of an arena this Basecaller keeps, so a call whose shapes repeat allocates nothing, and what a
        # call returns stays intact until TWO further calls have been issued on this Basecaller (device.Arena).  One Basecaller = one
        # stream of work: issue its calls in order on one stream.
        self._arena = None
        if borrow:
            from . import device as D
            self._arena = D.Arena(generations=2)
        if not isinstance(network, layers.Layer):
            raise TypeError("network must be a sloika_amd.layers.Layer")
        self.network = network
        self.kmer_len, self.nbase, self.min_prob, self.skip = kmer_len, nbase, min_prob, skip
        self.normalisation = normalisation
        self.in_flight = max(1, int(in_flight))
        if self.in_flight > 2:
            from . import device as D
            D.want_hw_queues(4 * self.in_flight)
        self.fused_decode = FUSED_DECODE if fused_decode is None else bool(fused_decode)
        self._ws = decode.ViterbiWorkspace()
        _lib.lib()
        try:                                   # one-time host work that does not belong into the first call (layers._cu_count)
            import torch
            if torch.cuda.is_available():
                layers._cu_count(torch.device("cuda", torch.cuda.current_device()))
        except (ImportError, RuntimeError):
            pass

    def _hidden(self, chunks, upto):
        """Run the network on [B, chunk_len] device signal up to (not including) layer index `upto`."""
        from . import device as D
        cd = D.to_dev(chunks)
        net = self.network
        seq = net.layers if isinstance(net, layers.Serial) else [net]
        first = seq[0]
        if cd.dim() == 3:
            # event-feature models (models/baseline_lstm.py, baseline_gru.py: Window over 4 features per event): the input is the
            # [T, B, features] tensor itself, as `calc_post` takes it (basecall.py:73-75); nothing to normalise
            if cd.shape[2] != first.insize:
                raise ValueError("feature input has %d features per step, the network takes %d" % (cd.shape[2], first.insize))
            x, rest = cd, seq[:upto]
        elif first.insize != 1:
            raise ValueError("this network takes %d features per step: hand over a [T, B, %d] feature tensor, not raw chunks"
                             % (first.insize, first.insize))
        elif (self.normalisation == 'per-chunk' and isinstance(first, layers.Convolution) and first.insize == 1
                and len(seq) > 1):
            # the conv front end reads the chunk-major normalised signal directly: no [T,B,1] transpose
            norm = batch.normalise_chunks(cd, 'per-chunk', out_layout='chunk')
            B, T = norm.shape
            x = first.run_strided(norm.data_ptr(), T, B, 1, T, norm.device)
            rest = seq[1:upto]
        else:
            x = batch.normalise_chunks(cd, self.normalisation, out_layout='network')
            rest = seq[:upto]
        keep = layers._HINTS.in_flight, layers._HINTS.deterministic       # (thread local: one forward pass per host thread at a time)
        layers._HINTS.in_flight, layers._HINTS.deterministic = self.in_flight, self.deterministic
        try:
            for layer in rest:
                x = layer._forward(x, None, False)
        finally:
            layers._HINTS.in_flight, layers._HINTS.deterministic = keep
        return x

    def posteriors(self, chunks):
        """[B, chunk_len] device signal -> [T', B, nstate] posteriors (network layout)."""
        net = self.network
        n = len(net.layers) if isinstance(net, layers.Serial) else 1
        return self._hidden(chunks, n)

    #: row widths csrc/softmax_viterbi.hip is instantiated for; any narrower Softmax input is decoded from rows with zero columns up to
    #: the next of them (free when the rows come out of a zero-padded Gru twin, otherwise one padded copy of the hidden state: 0.3 ms
    #: at B = 1024 against the 2 ms of the projection + logits decoder pair)
    FUSED_WIDTHS = (64, 96, 112, 128)

    def _fused_pack(self, last, hid):
        """(the Softmax layer's weights packed for csrc/softmax_viterbi.hip, the hidden state with the row width the kernel reads), or
        None when that kernel does not apply."""
        import torch
        if not self.fused_decode or hid.stride(2) != 1 or hid.stride(0) != hid.shape[1] * hid.stride(1):
            return None
        kp = next((k for k in self.FUSED_WIDTHS if k >= last.insize), None)
        if kp is None:
            return None
        if kp == last.insize:
            if hid.stride(1) % 4 or hid.data_ptr() % 16:
                return None
            pack = last.viterbi_pack(self.nbase, self.kmer_len)
            return None if pack is None else (pack, hid)
        pack = last.viterbi_pack(self.nbase, self.kmer_len, kpad=kp)
        if pack is None:
            return None
        if getattr(hid, "_slk_zero_padded", 0) >= kp and hid.stride(1) >= kp and hid.stride(1) % 4 == 0 and hid.data_ptr() % 16 == 0:
            # straight out of a zero-padded Gru twin (layers.Gru._forward): the columns behind insize exist and are zero
            return pack, hid.as_strided((hid.shape[0], hid.shape[1], kp), hid.stride())
        wide = torch.zeros((hid.shape[0], hid.shape[1], kp), dtype=hid.dtype, device=hid.device)
        wide[:, :, :last.insize] = hid
        return pack, wide

    def call_chunks(self, chunks, lp_dump=None):
        """-> device tensors (scores float32 [B], paths int32 [B, T'] (-1 padded), lens int32 [B]).

        When the network ends in a Softmax layer whose shape csrc/softmax_viterbi.hip covers, the decoder starts from that
        layer's INPUT and neither the logits nor the posterior (3.4 GB each at B=1024) are ever written; `lp_dump`, a float32
        device tensor [T', B, nstate], then receives the log-posteriors the dynamic programme consumed (tests).  Otherwise
        (and with fused_decode=False) the decoder consumes the layer's logits + row statistics, bit-identical to decoding
        `posteriors()`."""
        if self._arena is None:
            return self._call_chunks(chunks, lp_dump)
        with self._arena:
            return self._call_chunks(chunks, lp_dump)

    def _call_chunks(self, chunks, lp_dump):
        net = self.network
        last = net.layers[-1] if isinstance(net, layers.Serial) else None
        if type(last) is layers.Softmax and len(net.layers) > 1:
            hid = self._hidden(chunks, len(net.layers) - 1)
            packed = self._fused_pack(last, hid)
            if packed is not None:
                return decode.viterbi_fused_batch(packed[1], packed[0], self.kmer_len, skip_pen=self.skip, nbase=self.nbase,
                                                  min_prob=self.min_prob, workspace=self._ws, lp_dump=lp_dump)
            if lp_dump is not None:
                raise ValueError("lp_dump needs the fused decoder (csrc/softmax_viterbi.hip does not cover this network)")
            logits, stats, ld = last.logits_and_stats(hid)
            T, B = hid.shape[0], hid.shape[1]
            return decode.viterbi_logits_batch(logits, stats, self.kmer_len, T, B, ld=ld, skip_pen=self.skip,
                                               nbase=self.nbase, min_prob=self.min_prob, workspace=self._ws)
        post = self.posteriors(chunks)
        return decode.viterbi_batch(post, self.kmer_len, skip_pen=self.skip, nbase=self.nbase,
                                    min_prob=self.min_prob, workspace=self._ws)

    _BATCH_STREAMS = {}

    @classmethod
    def batch_slots(cls, network, in_flight=8, **kwargs):
        """The slots of call_batches, for reuse over several streams of batches: (Basecallers with arenas of their own, their pinned
        result buffers -- two sets per slot, like the arena's two device sets)."""
        nslot = max(1, int(in_flight))
        return [cls(network, in_flight=nslot, borrow=True, **kwargs) for _ in range(nslot)], [[None, None] for _ in range(nslot)]

    @classmethod
    def call_batches(cls, network, batches, in_flight=8, copy=True, slots=None, **kwargs):
        """A stream of batches with `in_flight` of them on the device at a time: generator over `batches` (an iterable of [B, chunk_len]
        signal batches -- device tensors, or host arrays that are uploaded -- or [T, B, features] tensors for event models), yielding
        (scores float32 [B], paths int32 [B, T'] (-1 padded), lens int32 [B]) as numpy arrays ON THE HOST, one per batch, in the order
        of the input.

        What the reference does with a pool of worker processes (bin/basecall_network.py:100-104, one read per call) a GPU does with
        batches side by side: at the north star's batch of 256 chunks one batch fills 64 of 256 CUs, so several must run at once.  The
        set of streams is FIXED: `in_flight` slots, each with a Basecaller(borrow=True) (a call allocates nothing), a stream (plus the
        side stream the directions of a birnn share their batch's stream with, layers.Parallel), pinned result buffers, and ONE copy
        stream; batch i runs in slot i % in_flight.  Once batch i is queued in its slot the generator hands out batch i - in_flight, the
        slot's batch before (waits for its copy; the slot's two result sets, on the device and on the host, take turns): the host is a
        consumer, a slot holds the batch that runs and at most one queued behind it, the device is never without work -- not even with
        one slot.  8 slots use 17
        streams: below the 32 hardware queues sloika_amd asks for (device.want_hw_queues), so no queue is time-sliced.

        copy=False yields views of the slot's pinned buffers instead of copies: valid until `in_flight` further batches have been
        yielded.  slots: what batch_slots(network, in_flight, ...) returned, for a caller that runs stream after stream (a server): the
        Basecallers keep their arenas and the pinned buffers between the streams, so a later stream allocates nothing at all."""
        import torch
        from . import device as D
        if slots is None:
            slots = cls.batch_slots(network, in_flight, **kwargs)
        bcs, host = slots
        nslot = len(bcs)
        D.want_hw_queues(2 * nslot + 1)
        # the streams are kept per device and slot count: torch hands out streams from a pool of 32 round robin, and a process that has
        # USED more streams than the runtime has hardware queues (32) gets every queue time-sliced (measured: baseline_raw_gru, eight in
        # flight, 640 M samples/s with 17 streams used, 150-210 M once another leg's 17 had been used before)
        key = (torch.cuda.current_device(), nslot)
        if key not in cls._BATCH_STREAMS:
            cls._BATCH_STREAMS[key] = ([torch.cuda.Stream() for _ in range(nslot)], torch.cuda.Stream())
        streams, copy_stream = cls._BATCH_STREAMS[key]
        pending = [None] * nslot
        ncall = [0] * nslot

        def collect(p):
            ev, bufs, B, T = p
            ev.synchronize()
            sc, pa, le = bufs[0][:B].numpy(), bufs[1][:B, :T].numpy(), bufs[2][:B].numpy()
            return (sc.copy(), pa.copy(), le.copy()) if copy else (sc, pa, le)

        for i, chunks in enumerate(batches):
            k = i % nslot
            before = pending[k]                            # batch i - in_flight: handed out once batch i is queued behind it
            cur = torch.cuda.current_stream()
            s = streams[k]
            s.wait_stream(cur)                             # (whatever produced the batch on the caller's stream)
            with torch.cuda.stream(s):
                cd = D.to_dev(chunks)
                if isinstance(chunks, torch.Tensor) and chunks.is_cuda:
                    chunks.record_stream(s)
                scores, paths, lens = bcs[k].call_chunks(cd)
                done = torch.cuda.Event()
                done.record(s)
            B, T = paths.shape
            g = ncall[k] & 1
            ncall[k] += 1
            bufs = host[k][g]
            if bufs is None or bufs[1].shape[0] < B or bufs[1].shape[1] < T:
                bufs = host[k][g] = (torch.empty((B,), dtype=torch.float32).pin_memory(),
                                     torch.empty((B, T), dtype=torch.int32).pin_memory(),
                                     torch.empty((B,), dtype=torch.int32).pin_memory())
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done)
                bufs[0][:B].copy_(scores, non_blocking=True)
                bufs[1][:B, :T].copy_(paths, non_blocking=True)
                bufs[2][:B].copy_(lens, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            pending[k] = (ev, bufs, B, T)
            if before is not None:
                yield collect(before)
        # the batches still on the device, oldest first
        n = sum(ncall)
        for i in range(n - min(n, nslot), n):
            if pending[i % nslot] is not None:
                yield collect(pending[i % nslot])
                pending[i % nslot] = None

    def call_bases(self, chunks, alphabet='ACGT'):
        """call_chunks + states -> bases on the device (what basecall.SeqPrinter.write does per read, basecall.py:157-163,
        with always_move as for a transducer model): -> (scores device [B], list of B base strings)."""
        from . import bio
        scores, paths, lens = self.call_chunks(chunks)
        return scores, bio.paths_to_bases(paths, lens, self.kmer_len, alphabet, always_move=True)

    def _trim_reads(self, signals, trim, open_pore_fraction):
        """basecall.py:111-112: trim_open_pore (which also cuts the read to whole 100-sample windows), then trim_array."""
        from . import util
        sigs = batch.trim_open_pore_many(signals, open_pore_fraction)
        sigs = [util.trim_array(s, *trim) for s in sigs]
        if min(len(s) for s in sigs) < 1:
            raise ValueError("empty read after trimming")
        return sigs

    @staticmethod
    def _pack_reads(sigs):
        """Trimmed reads (host float32 arrays) -> zero-padded [B, Lmax] device tensor (through pinned memory when it is big)."""
        import torch
        from . import device as D
        nsamp = [len(s) for s in sigs]
        B, lmax = len(sigs), max(nsamp)
        host = torch.zeros((B, lmax), dtype=torch.float32)
        if B * lmax >= (1 << 20):
            host = host.pin_memory()
        hv = host.numpy()
        for b, sig in enumerate(sigs):
            hv[b, :nsamp[b]] = sig
        return host.to(D.device(), non_blocking=True), nsamp

    def _call_padded(self, padded, nsamp):
        """One padded batch of trimmed reads resident on the device ([B, Lmax], read b in its first nsamp[b] samples): per-read
        normalisation, network and decoder with per-read lengths.  -> (scores, paths, lens) on the device."""
        if self._arena is None:
            return self._call_padded_pass(padded, nsamp)
        with self._arena:
            return self._call_padded_pass(padded, nsamp)

    def _call_padded_pass(self, padded, nsamp):
        net = self.network
        B = padded.shape[0]
        keep = layers._HINTS.in_flight, layers._HINTS.deterministic
        # ragged batches side by side keep the four-chunk plan: their workgroups queue for the CUs and a batch of short reads hands its
        # CUs on early, which a plan that packs more chunks into fewer, slower workgroups would not let it do
        layers._HINTS.in_flight, layers._HINTS.deterministic = 1, self.deterministic
        try:
            with layers.ragged(nsamp) as ctx:
                x = batch.normalise_reads_ragged(padded, ctx.lengths)      # per-read normalisation (basecall.py:117-118)
                hid = x
                for layer in net.layers[:-1]:
                    hid = layer._forward(hid, None, False)
                lengths = layers.ragged.current
                packed = self._fused_pack(net.layers[-1], hid)
                pack = packed[0] if packed is not None else None
                if pack is None:
                    logits, stats, ld = net.layers[-1].logits_and_stats(hid)
                else:
                    hid = packed[1]
        finally:
            layers._HINTS.in_flight, layers._HINTS.deterministic = keep
        T = hid.shape[0]
        if pack is not None:
            return decode.viterbi_fused_batch(hid, pack, self.kmer_len, skip_pen=self.skip, nbase=self.nbase,
                                              min_prob=self.min_prob, workspace=self._ws, lengths=lengths.contiguous())
        return decode.viterbi_logits_batch(logits, stats, self.kmer_len, T, B, ld=ld, skip_pen=self.skip, nbase=self.nbase,
                                           min_prob=self.min_prob, workspace=self._ws, lengths=lengths.contiguous())

    def _call_trimmed(self, sigs):
        padded, nsamp = self._pack_reads(sigs)
        return self._call_padded(padded, nsamp)

    def call_reads(self, signals, trim=(0, 0), open_pore_fraction=0.0):
        """Whole reads of different lengths in ONE batch (the reference calls them one at a time, basecall.py:88-121):
        `signals` is a list of 1-D float arrays (already scaled, e.g. fast5.Fast5.get_read()); each is trimmed as raw_worker does, median/MAD
        normalised over its own length, zero-padded to the longest, and the network + decoder run on the padded batch
        with per-read lengths (layers.ragged), so every read gets exactly what a batch-1 call would give.
        -> device tensors (scores [B], paths [B, T'max] (-1 padded), lens [B]) and the per-read sample counts."""
        net = self.network
        if not isinstance(net, layers.Serial) or type(net.layers[-1]) is not layers.Softmax:
            raise ValueError("call_reads needs a Serial network ending in a Softmax layer")
        sigs = self._trim_reads(signals, trim, open_pore_fraction)
        scores, paths, lens = self._call_trimmed(sigs)
        return scores, paths, lens, [len(s) for s in sigs]

    @staticmethod
    def length_buckets(nsamp, max_batch=256, max_waste=0.08):
        """Group read indices into batches of similar length: reads sorted by length, a batch closed when it holds `max_batch`
        reads or when padding every member to the longest would waste more than `max_waste` of the batch's steps.  -> list of
        index lists (longest reads first: the big batches start while the host still packs the small ones)."""
        order = sorted(range(len(nsamp)), key=lambda i: -nsamp[i])
        buckets, cur = [], []
        for i in order:
            if cur:
                lmax = nsamp[cur[0]]
                used = sum(nsamp[j] for j in cur) + nsamp[i]
                if len(cur) >= max_batch or 1.0 - used / float(lmax * (len(cur) + 1)) > max_waste:
                    buckets.append(cur)
                    cur = []
            cur.append(i)
        if cur:
            buckets.append(cur)
        return buckets

    @classmethod
    def prepare_read_batches(cls, network, signals, trim=(0, 0), open_pore_fraction=0.0, max_batch=256, max_waste=0.08, ids=None,
                             **kwargs):
        """Preparation of the whole-read mode: the read set goes to the device in one upload, trimming bounds come from one launch over
        all windows (basecall.py:111-112), reads are bucketed by length and every bucket becomes a zero-padded device batch (one launch
        per bucket).  -> (batches, nsamp): batches = [(read indices, padded device tensor [B, Lmax], their sample counts)], nsamp =
        sample count of every read after trimming.

        A read that cannot be called -- no sample left after trimming, shorter than one open-pore window, no window livelier than
        the threshold, or a sample that is not finite -- is left out of every batch and gets nsamp 0; `failed_reads(nsamp)` lists
        them.  The reference's worker does the same one read at a time: it reports the read on stderr, returns None and the pool goes
        on (basecall.py:103-115).  The other reads of the set are unaffected."""
        import sys
        import torch
        from . import device as D
        # ONE upload of the whole read set; trimming bounds from the device's window spreads; the padded batches are then built on
        # the device (a launch per bucket) -- the host touches every sample once
        dev, off, lens = batch.upload_reads_windowed(signals)
        bad = batch.reads_nonfinite(dev, off, lens)
        bounds = batch.open_pore_bounds_many(dev, off, lens, open_pore_fraction)
        assert trim[0] >= 0 and trim[1] >= 0
        spans, nsamp = [], []
        for r, bd in enumerate(bounds):
            lo, hi = (0, 0) if (bd is None or bad[r]) else (bd[0] + trim[0], bd[1] - trim[1])               # util.trim_array
            spans.append((lo, hi))
            nsamp.append(max(0, hi - lo))
            if nsamp[-1] < 1:
                why = "samples that are not finite" if bad[r] else ("too short to trim the open pore" if bd is None else
                                                                    "nothing left after trimming")
                sys.stderr.write("Failure calling read {}: {}\n".format(r if ids is None else ids[r], why))
        good = [r for r in range(len(nsamp)) if nsamp[r] > 0]
        batches = []
        L = _lib.lib()
        for sub in cls.length_buckets([nsamp[r] for r in good], max_batch, max_waste):
            idx = [good[j] for j in sub]
            ns = [nsamp[i] for i in idx]
            padded = torch.empty((len(idx), max(ns)), dtype=torch.float32, device=dev.device)
            start = torch.as_tensor(np.asarray([int(off[i]) + spans[i][0] for i in idx], dtype=np.int64)).to(dev.device)
            ln = torch.as_tensor(np.asarray(ns, dtype=np.int32)).to(dev.device)
            _lib.check(L.slk_pack_reads_f32(dev.data_ptr(), start.data_ptr(), ln.data_ptr(), len(idx), padded.data_ptr(),
                                            padded.shape[1], D.stream_ptr()), "pack_reads")
            batches.append((idx, padded, ns))
        return batches, nsamp

    @staticmethod
    def failed_reads(nsamp):
        """Indices of the reads prepare_read_batches left out (their sample count after trimming is 0)."""
        return [i for i, n in enumerate(nsamp) if n < 1]

    @classmethod
    def run_read_batches(cls, network, batches, nreads, in_flight=None, lanes=None, **kwargs):
        """Device side: every prepared batch through normalisation, network and decoder, the batches spread over `in_flight`
        streams (default: one per batch, at most 8 -- a batch of 200 long reads occupies 50 of the 256 CUs for tens of
        milliseconds, so the chip only fills up with several of them side by side).  `lanes`: a list of (Basecaller, stream)
        pairs to reuse (read_lanes(); torch's allocator caches device memory per stream, so a server that keeps its lanes does
        not pay for gigabytes of fresh allocations on every call).  -> (scores [N] float32, list of N int32 path arrays) on the
        host; a read that is in no batch (failed_reads) has score NaN and path None."""
        if lanes is None:
            lanes = cls.read_lanes(network, max(1, min(8, len(batches)) if in_flight is None else in_flight), **kwargs)
        scores = np.full(nreads, np.nan, dtype=np.float32)
        paths = [None] * nreads
        cls._collect_read_batches(cls._launch_read_batches(lanes, batches), scores, paths)
        return scores, paths

    @staticmethod
    def _launch_read_batches(lanes, batches, first_lane=0):
        """Queue every prepared batch on its lane (batch k on lane (first_lane + k) % lanes) and its results' way to the host behind it;
        nothing here waits for the device.  -> pending [(read indices, host tensors, event, device tensors)]."""
        import torch
        nfl = len(lanes)
        cur = torch.cuda.current_stream()
        pending = []
        for k, (idx, padded, ns) in enumerate(batches):
            bc, s = lanes[(first_lane + k) % nfl]
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                padded.record_stream(s)
                res = bc._call_padded(padded, ns)
                host = tuple(t.to("cpu", non_blocking=True) for t in res)
                ev = torch.cuda.Event()
                ev.record(s)
            pending.append((idx, host, ev, res))
        return pending

    @staticmethod
    def _collect_read_batches(pending, scores, paths, ids=None):
        """Wait for the queued batches and file their results under the reads' indices (`ids`: slice-local index -> index in the set)."""
        for idx, host, ev, res in pending:
            ev.synchronize()
            sc, pa, le = (h.numpy() for h in host)
            for j, i in enumerate(idx):
                g = i if ids is None else ids[i]
                scores[g] = sc[j]
                paths[g] = pa[j, :le[j]].copy()

    @classmethod
    def read_lanes(cls, network, n, **kwargs):
        """n (Basecaller, stream) pairs sharing one network, for run_read_batches / call_reads_bucketed."""
        import torch
        return [(cls(network, in_flight=n, **kwargs), torch.cuda.Stream()) for _ in range(max(1, n))]

    @classmethod
    def call_reads_bucketed(cls, network, signals, trim=(0, 0), open_pore_fraction=0.0, max_batch=256, max_waste=0.08, in_flight=None,
                            lanes=None, stream_buckets=True, **kwargs):
        """Whole-read mode for MANY reads (what bin/basecall_network.py does with a pool of workers, basecall_network.py:100-104):
        reads are bucketed by length (length_buckets), every bucket is one padded ragged batch, and the buckets run side by side
        on streams of their own (one Basecaller each, sharing the network).  Each read gets bit for bit what call_reads([read])
        gives.  -> (scores [N] float32, list of N int32 path arrays, sample counts [N], stats) all on the host; stats holds the
        padded-step waste and the indices of the reads that could not be called (`failed`: score NaN, path None, sample count 0 --
        each reported on stderr as the reference's worker reports a read it skips, basecall.py:103-115)."""
        if open_pore_fraction == 0 and len(signals) > 2 * max_batch and stream_buckets:
            return cls._call_reads_streamed(network, signals, trim, max_batch, max_waste, in_flight, lanes, **kwargs)
        batches, nsamp = cls.prepare_read_batches(network, signals, trim, open_pore_fraction, max_batch, max_waste, **kwargs)
        scores, paths = cls.run_read_batches(network, batches, len(nsamp), in_flight, lanes, **kwargs)
        used = sum(nsamp)
        padded = sum(ns[0] * len(idx) for idx, _, ns in batches)
        stats = {"reads": len(nsamp), "batches": len(batches), "samples": used, "padded_samples": padded,
                 "padded_step_waste": 1.0 - used / float(max(padded, 1)), "failed": cls.failed_reads(nsamp)}
        return scores, paths, nsamp, stats

    _UPLOAD_STREAMS = {}

    @classmethod
    def _call_reads_streamed(cls, network, signals, trim, max_batch, max_waste, in_flight, lanes, window_size=100, **kwargs):
        """call_reads_bucketed for a big set with the CLI's open-pore fraction of 0 (bin/basecall_network.py:71), WITHOUT a round trip to
        the host between the upload of a read and its call: reads are bucketed by their RAW lengths (what the host knows without touching
        a sample; trimming takes at most a few windows off); bucket after bucket the host packs the reads into pinned memory and queues
        the upload on a copy stream, and the bucket's lane does the rest in stream order -- the check for samples that are not finite,
        the window spreads, trim_open_pore + trim_array on the device (slk_open_pore_trim_f32), the zero-padded batch, normalisation,
        network, decoder, results and trimmed lengths to the host.  The host packs bucket k + 1 while the device runs bucket k: what is
        left in front of the network is the first bucket's upload (round 5: 85 ms of packing, upload and trimming for 4096 reads before
        the first network kernel).  A read that fails keeps its place in its bucket with length 0 (columns of a batch never mix), is
        reported on stderr like the reference's worker does (basecall.py:103-115) and comes back with score NaN and path None."""
        import sys
        import torch
        from . import device as D
        L = _lib.lib()
        dev = D.device()
        nread = len(signals)
        raw = [len(s) for s in signals]
        buckets = cls.length_buckets(raw, max_batch, max_waste)
        if lanes is None:
            lanes = cls.read_lanes(network, max(1, min(8, len(buckets))) if in_flight is None else in_flight, **kwargs)
        up = cls._UPLOAD_STREAMS.get(dev.index)
        if up is None:
            up = cls._UPLOAD_STREAMS[dev.index] = torch.cuda.Stream()
        # one pinned staging area for the whole set (grow-only, kept per host thread: batch._staging), bucket after bucket
        strides = [-(-n // window_size) * window_size for n in raw]
        bsize = [sum(strides[i] for i in idx) for idx in buckets]
        total = sum(bsize)
        st = batch._staging
        if getattr(st, "buf", None) is None or st.buf.numel() < total:
            st.buf = torch.empty(max(total, 1 << 20), dtype=torch.float32).pin_memory()
            st.event = None
        if st.event is not None:
            st.event.synchronize()
        hv = st.buf.numpy()
        assert trim[0] >= 0 and trim[1] >= 0
        import concurrent.futures
        pool = concurrent.futures.ThreadPoolExecutor(min(8, os.cpu_count() or 1))
        pending, base = [], 0
        try:
            for k, idx in enumerate(buckets):
                n = len(idx)
                off = np.concatenate([[0], np.cumsum([strides[i] for i in idx])]).astype(np.int64)

                def pack(lo, hi, idx=idx, off=off, base=base):
                    for j in range(lo, hi):
                        i = idx[j]
                        hv[base + off[j]: base + off[j] + raw[i]] = signals[i]
                        hv[base + off[j] + raw[i]: base + off[j + 1]] = 0.0
                step = max(1, -(-n // 8))
                list(pool.map(lambda a: pack(*a), [(lo, min(n, lo + step)) for lo in range(0, n, step)]))
                # [first sample | first window | raw length | whole windows] of every read, one small upload
                meta = torch.empty((4, n), dtype=torch.int64).pin_memory()
                mv = meta.numpy()
                mv[0], mv[1], mv[2], mv[3] = off[:n], off[:n] // window_size, [raw[i] for i in idx], [raw[i] // window_size for i in idx]
                with torch.cuda.stream(up):
                    sig = st.buf[base: base + bsize[k]].to(dev, non_blocking=True)
                    meta_d = meta.to(dev, non_blocking=True)
                    uploaded = torch.cuda.Event()
                    uploaded.record(up)
                base += bsize[k]
                bc, lane = lanes[k % len(lanes)]
                with torch.cuda.stream(lane):
                    lane.wait_event(uploaded)
                    sig.record_stream(lane)
                    meta_d.record_stream(lane)
                    first_sample, first_win = meta_d[0], meta_d[1]
                    rawlen, nwin = meta_d[2].to(torch.int32), meta_d[3].to(torch.int32)
                    flags = torch.zeros((n,), dtype=torch.int32, device=dev)
                    _lib.check(L.slk_reads_nonfinite_f32(sig.data_ptr(), first_sample.data_ptr(), rawlen.data_ptr(), n,
                                                         int(max(raw[i] for i in idx)), flags.data_ptr(), lane.cuda_stream), "reads_nonfinite")
                    _, _, spread = batch.normalise_chunks(sig.view(-1, window_size), 'per-chunk', return_stats=True)
                    start = torch.empty((n,), dtype=torch.int64, device=dev)
                    ln = torch.empty((n,), dtype=torch.int32, device=dev)
                    _lib.check(L.slk_open_pore_trim_f32(spread.data_ptr(), first_win.data_ptr(), nwin.data_ptr(), first_sample.data_ptr(),
                                                        n, window_size, int(trim[0]), int(trim[1]), start.data_ptr(), ln.data_ptr(),
                                                        flags.data_ptr(), lane.cuda_stream), "open_pore_trim")
                    lmax = max(raw[i] for i in idx)                    # (an upper bound of the trimmed lengths: the batch's row width)
                    padded = torch.empty((n, lmax), dtype=torch.float32, device=dev)
                    _lib.check(L.slk_pack_reads_f32(sig.data_ptr(), start.data_ptr(), ln.data_ptr(), n, padded.data_ptr(), lmax,
                                                    lane.cuda_stream), "pack_reads")
                    # a failed read (length 0) runs as one zero sample: its column is garbage nobody reads
                    res = bc._call_padded(padded, ln.clamp(min=1))
                    host = tuple(t.to("cpu", non_blocking=True) for t in res + (ln, flags))
                    ev = torch.cuda.Event()
                    ev.record(lane)
                pending.append((idx, host, ev, res, lmax))
            st.event = torch.cuda.Event()
            st.event.record(up)
        finally:
            pool.shutdown(wait=True)
        scores = np.full(nread, np.nan, dtype=np.float32)
        paths = [None] * nread
        nsamp = [0] * nread
        padded_total = 0
        for idx, host, ev, res, lmax in pending:
            ev.synchronize()
            sc, pa, le, ns, fl = (h.numpy() for h in host)
            padded_total += lmax * len(idx)
            for j, i in enumerate(idx):
                if fl[j]:
                    why = "samples that are not finite" if fl[j] & 1 else ("too short to trim the open pore" if fl[j] & 2 else
                                                                            "nothing left after trimming")
                    sys.stderr.write("Failure calling read {}: {}\n".format(i, why))
                    continue
                nsamp[i] = int(ns[j])
                scores[i] = sc[j]
                paths[i] = pa[j, :le[j]].copy()
        used = sum(nsamp)
        stats = {"reads": nread, "batches": len(buckets), "samples": used, "padded_samples": padded_total,
                 "padded_step_waste": 1.0 - used / float(max(padded_total, 1)), "failed": cls.failed_reads(nsamp), "streamed": True}
        return scores, paths, nsamp, stats

    def call_chunks_host(self, chunks):
        scores, paths, lens = self.call_chunks(chunks)
        scores, paths, lens = scores.cpu().numpy(), paths.cpu().numpy(), lens.cpu().numpy()
        return scores, [paths[i, : lens[i]].tolist() for i in range(len(lens))]


def synthetic_chunks(nchunk, chunk_len=4000, seed=0xdeadbeef, dwell=10.0, noise=0.15, first_chunk=0):
    """Synthetic raw signal: per chunk, piecewise-constant levels ~N(0,1) with geometric dwell
    (mean `dwell` samples) plus N(0, noise^2), RandomState(seed + chunk_id), scaled to a pA-like range so that
    normalisation does real work.  float32 [nchunk, chunk_len]."""
    out = np.empty((nchunk, chunk_len), dtype=np.float32)
    for c in range(nchunk):
        rs = np.random.RandomState((seed + first_chunk + c) % (2 ** 32))
        nseg = int(chunk_len / dwell * 2) + 16
        d = rs.geometric(1.0 / dwell, size=nseg)
        while d.sum() < chunk_len:
            d = np.concatenate([d, rs.geometric(1.0 / dwell, size=nseg)])
        levels = rs.normal(size=len(d))
        sig = np.repeat(levels, d)[:chunk_len]
        sig = sig + rs.normal(scale=noise, size=chunk_len)
        out[c] = (sig * 12.0 + 90.0).astype(np.float32)
    return out
