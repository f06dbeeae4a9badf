"""Minimal fast5 (HDF5) reader for the input side of the basecalling path.

The reference reads its inputs through `fast5_research.Fast5` (requirements.txt:5; call sites sloika/basecall.py:103-109,
sloika/batch.py:115-140), which sits on h5py/libhdf5 -- neither exists on the build or the GPU box.  ONT's single-read
fast5 files are plain HDF5 1.8 files of a very regular shape (superblock v0, old-style groups, version-1 object headers,
chunked + deflated integer datasets, a handful of scalar attributes), so this module parses exactly that subset with the
standard library: enough for `Fast5(path).get_read(raw=True)` (the scaled current the network is fed), the strand
summary the workers use, and the basecall ONT's own software left in the file (used as an accuracy yardstick in tests).

Format facts follow the public HDF5 File Format Specification 2.0 (sections cited inline).  Unsupported constructs raise
`Fast5Error` rather than guess.
"""
import struct
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Fast5Error(IOError):
    pass


class _Dataset(object):
    def __init__(self, f, msgs):
        self.f, self.msgs = f, msgs

    @property
    def attrs(self):
        return self.f._attrs(self.msgs)


class _Group(object):
    def __init__(self, f, msgs):
        self.f, self.msgs = f, msgs
        self._links = None

    @property
    def attrs(self):
        return self.f._attrs(self.msgs)

    def links(self):
        if self._links is None:
            self._links = {}
            for mtype, body in self.msgs:
                if mtype == 0x0011:                         # Symbol Table message (IV.A.2.r): B-tree + local heap
                    btree, heap = struct.unpack_from("<QQ", body, 0)
                    self.f._walk_group_btree(btree, self.f._local_heap(heap), self._links)
        return self._links

    def keys(self):
        return sorted(self.links())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, _Group) or part not in node.links():
                raise KeyError(path)
            node = node.f._object(node.links()[part])
        return node


class HDF5File(object):
    """Read-only view of an HDF5 file with a version-0/1 superblock and old-style groups."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        if self.buf[:8] != _SIG:
            raise Fast5Error("%s: not an HDF5 file" % path)
        ver = self.buf[8]
        if ver not in (0, 1):
            raise Fast5Error("%s: superblock version %d is not supported (0/1 only)" % (path, ver))
        so, sl = self.buf[13], self.buf[14]
        if (so, sl) != (8, 8):
            raise Fast5Error("only 8-byte offsets/lengths are supported")
        off = 24 if ver == 0 else 28                       # (II.A) v1 adds indexed-storage K + reserved
        self.base = struct.unpack_from("<Q", self.buf, off)[0]
        root_ste = off + 32                                 # base, free-space, EOF, driver-info addresses precede it
        self.root = self._object(struct.unpack_from("<Q", self.buf, root_ste + 8)[0])

    # ---- object headers (IV.A.1.a, version 1) ----
    def _messages(self, addr):
        buf, a = self.buf, addr + self.base
        if buf[a] != 1:
            raise Fast5Error("object header version %d is not supported" % buf[a])
        nmsg, = struct.unpack_from("<H", buf, a + 2)
        hsize, = struct.unpack_from("<I", buf, a + 8)
        blocks = [(a + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", buf, pos)
                body = buf[pos + 8: pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:                          # continuation (IV.A.2.q)
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((caddr + self.base, clen))
                out.append((mtype, body))
        return out

    def _object(self, addr):
        msgs = self._messages(addr)
        types = {m for m, _ in msgs}
        return _Group(self, msgs) if 0x0011 in types else _Dataset(self, msgs)

    # ---- groups: local heap (III.D) and version-1 B-tree of symbol nodes (III.A.1, III.C) ----
    def _local_heap(self, addr):
        a = addr + self.base
        if self.buf[a:a + 4] != b"HEAP":
            raise Fast5Error("bad local heap signature")
        size, _free, data = struct.unpack_from("<QQQ", self.buf, a + 8)
        return self.buf[data + self.base: data + self.base + size]

    def _walk_group_btree(self, addr, heap, links):
        a = addr + self.base
        if self.buf[a:a + 4] != b"TREE":
            raise Fast5Error("bad B-tree signature")
        ntype, level, used = struct.unpack_from("<BBH", self.buf, a + 4)
        if ntype != 0:
            raise Fast5Error("expected a group B-tree")
        pos = a + 24                                         # after left/right sibling addresses
        for i in range(used):
            child, = struct.unpack_from("<Q", self.buf, pos + 8 + 16 * i)     # key_i (8) then child_i (8)
            if level > 0:
                self._walk_group_btree(child, heap, links)
            else:
                self._symbol_node(child, heap, links)

    def _symbol_node(self, addr, heap, links):
        a = addr + self.base
        if self.buf[a:a + 4] != b"SNOD":
            raise Fast5Error("bad symbol node signature")
        nsym, = struct.unpack_from("<H", self.buf, a + 6)
        for i in range(nsym):
            e = a + 8 + 40 * i
            name_off, ohdr = struct.unpack_from("<QQ", self.buf, e)
            end = heap.index(b"\0", name_off)
            links[heap[name_off:end].decode("utf-8")] = ohdr

    # ---- datatypes (IV.A.2.d), dataspaces (IV.A.2.b) ----
    def _datatype(self, body, pos=0):
        """-> (kind, numpy dtype or None, size in bytes, bytes consumed)."""
        cv = body[pos]
        cls, bits0 = cv & 0x0F, body[pos + 1]
        size, = struct.unpack_from("<I", body, pos + 4)
        order = ">" if (bits0 & 1) else "<"
        if cls == 0:                                         # fixed point
            signed = bool(bits0 & 0x08)
            return "int", np.dtype("%s%s%d" % (order, "i" if signed else "u", size)), size, 8 + 4
        if cls == 1:                                         # floating point
            return "float", np.dtype("%sf%d" % (order, size)), size, 8 + 12
        if cls == 3:                                         # fixed-length string
            return "string", None, size, 8
        if cls == 9:                                         # variable length (sequence or string)
            kind = "vlen_string" if (bits0 & 0x0F) == 1 else "vlen"
            _k, _d, _s, used = self._datatype(body, pos + 8)
            return kind, None, size, 8 + used
        raise Fast5Error("datatype class %d is not supported" % cls)

    @staticmethod
    def _dataspace(body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1:
            pos = 8
        elif ver == 2:
            pos = 4
        else:
            raise Fast5Error("dataspace version %d is not supported" % ver)
        dims = struct.unpack_from("<%dQ" % rank, body, pos) if rank else ()
        return tuple(int(d) for d in dims)

    def _vlen_bytes(self, ref):
        """Global heap object (III.E) behind a variable-length element: length(4), collection address(8), index(4)."""
        length, caddr, index = struct.unpack_from("<IQI", ref, 0)
        a = caddr + self.base
        if self.buf[a:a + 4] != b"GCOL":
            raise Fast5Error("bad global heap signature")
        csize, = struct.unpack_from("<Q", self.buf, a + 8)
        pos, end = a + 16, a + csize
        while pos + 16 <= end:
            idx, _ref, _res, osize = struct.unpack_from("<HHIQ", self.buf, pos)
            if idx == 0:
                break
            if idx == index:
                return self.buf[pos + 16: pos + 16 + osize][:length]
            pos += 16 + ((osize + 7) // 8) * 8
        raise Fast5Error("global heap object %d not found" % index)

    def _decode(self, kind, dtype, size, dims, raw):
        n = int(np.prod(dims)) if dims else 1
        if kind in ("int", "float"):
            arr = np.frombuffer(raw, dtype=dtype, count=n)
            return arr.reshape(dims) if dims else arr[0]
        if kind == "string":
            vals = [raw[i * size:(i + 1) * size].split(b"\0")[0].decode("utf-8", "replace") for i in range(n)]
        elif kind == "vlen_string":
            vals = [self._vlen_bytes(raw[i * size:(i + 1) * size]).decode("utf-8", "replace") for i in range(n)]
        else:
            raise Fast5Error("cannot decode %s data" % kind)
        return vals if dims else vals[0]

    # ---- attributes (IV.A.2.m, version 1) ----
    def _attrs(self, msgs):
        out = {}
        for mtype, body in msgs:
            if mtype != 0x000C:
                continue
            ver = body[0]
            if ver != 1:
                raise Fast5Error("attribute message version %d is not supported" % ver)
            nsize, tsize, ssize = struct.unpack_from("<HHH", body, 2)
            pad = lambda v: (v + 7) // 8 * 8
            pos = 8
            name = body[pos:pos + nsize].split(b"\0")[0].decode("utf-8")
            pos += pad(nsize)
            kind, dtype, size, _ = self._datatype(body, pos)
            pos += pad(tsize)
            dims = self._dataspace(body[pos:pos + ssize]) if ssize else ()
            pos += pad(ssize)
            out[name] = self._decode(kind, dtype, size, dims, body[pos:])
        return out

    # ---- dataset payload: data layout v3 (IV.A.2.i), filter pipeline v1 (IV.A.2.l), chunk B-tree (III.A.1) ----
    def read(self, ds):
        if not isinstance(ds, _Dataset):
            raise Fast5Error("not a dataset")
        msgs = dict((m, b) for m, b in ds.msgs if m in (0x0001, 0x0003, 0x0008, 0x000B))
        kind, dtype, size, _ = self._datatype(msgs[0x0003])
        dims = self._dataspace(msgs[0x0001])
        lay = msgs[0x0008]
        if lay[0] != 3:
            raise Fast5Error("data layout version %d is not supported" % lay[0])
        cls = lay[1]
        total = (int(np.prod(dims)) if dims else 1) * size
        if cls == 0:                                         # compact
            n, = struct.unpack_from("<H", lay, 2)
            raw = lay[4:4 + n]
        elif cls == 1:                                       # contiguous
            addr, n = struct.unpack_from("<QQ", lay, 2)
            raw = b"" if addr == _UNDEF else self.buf[addr + self.base: addr + self.base + n]
        elif cls == 2:                                       # chunked
            rank = lay[2]                                    # dataset rank + 1
            btree, = struct.unpack_from("<Q", lay, 3)
            cdims = struct.unpack_from("<%dI" % rank, lay, 11)
            filters = self._filters(msgs.get(0x000B))
            if rank != 2:
                raise Fast5Error("only one-dimensional chunked datasets are supported")
            out = bytearray(total)
            chunks = []
            if btree != _UNDEF:
                self._walk_chunk_btree(btree, rank, chunks)
            for csize, mask, offs, caddr in chunks:
                data = self.buf[caddr + self.base: caddr + self.base + csize]
                for k, (fid, _cd) in reversed(list(enumerate(filters))):
                    if mask & (1 << k):
                        continue                             # this filter was skipped for the chunk
                    if fid == 1:
                        data = zlib.decompress(data)
                    elif fid == 2:                           # shuffle: de-interleave bytes
                        a = np.frombuffer(data, dtype=np.uint8)
                        data = a.reshape(size, -1).T.tobytes() if len(a) % size == 0 else data
                    elif fid == 3:                           # fletcher32 checksum trails the chunk
                        data = data[:-4]
                    else:
                        raise Fast5Error("filter %d is not supported" % fid)
                start = offs[0] * size
                out[start:start + min(len(data), total - start)] = data[:max(0, total - start)]
            raw = bytes(out)
        else:
            raise Fast5Error("data layout class %d is not supported" % cls)
        return self._decode(kind, dtype, size, dims, raw)

    @staticmethod
    def _filters(body):
        if body is None:
            return []
        if body[0] != 1:
            raise Fast5Error("filter pipeline version %d is not supported" % body[0])
        n, pos, out = body[1], 8, []
        for _ in range(n):
            fid, nlen, _flags, ncd = struct.unpack_from("<HHHH", body, pos)
            pos += 8 + (nlen + 7) // 8 * 8
            cd = struct.unpack_from("<%dI" % ncd, body, pos)
            pos += 4 * ncd + (4 if ncd % 2 else 0)
            out.append((fid, cd))
        return out

    def _walk_chunk_btree(self, addr, rank, chunks):
        a = addr + self.base
        if self.buf[a:a + 4] != b"TREE":
            raise Fast5Error("bad chunk B-tree signature")
        ntype, level, used = struct.unpack_from("<BBH", self.buf, a + 4)
        if ntype != 1:
            raise Fast5Error("expected a chunk B-tree")
        ksize = 8 + 8 * rank
        pos = a + 24
        for i in range(used):
            k = pos + i * (ksize + 8)
            csize, mask = struct.unpack_from("<II", self.buf, k)
            offs = struct.unpack_from("<%dQ" % rank, self.buf, k + 8)
            child, = struct.unpack_from("<Q", self.buf, k + ksize)
            if level > 0:
                self._walk_chunk_btree(child, rank, chunks)
            else:
                chunks.append((csize, mask, offs, child))


class Fast5(object):
    """The slice of `fast5_research.Fast5` the basecalling path uses (sloika/basecall.py:103-109)."""

    def __init__(self, path):
        self.path = path
        self.h5 = HDF5File(path)
        ch = self.h5.root["UniqueGlobalKey/channel_id"].attrs
        self.channel_meta = {k: ch[k] for k in ("digitisation", "offset", "range", "sampling_rate") if k in ch}
        self.sample_rate = float(ch["sampling_rate"])

    def read_names(self):
        return self.h5.root["Raw/Reads"].keys()

    def _read_group(self):
        names = self.read_names()
        if not names:
            raise Fast5Error("%s holds no raw read" % self.path)
        return self.h5.root["Raw/Reads/" + names[0]]

    def get_read(self, raw=True, scale=True):
        """Raw signal of the (single) read: picoamperes `(adc + offset) * range / digitisation` as float, or the int16
        ADC values with scale=False."""
        if not raw:
            raise Fast5Error("only raw reads are supported (event tables belong to the deprecated event path)")
        adc = self.h5.read(self._read_group()["Signal"])
        if not scale:
            return adc
        m = self.channel_meta
        return (adc.astype(np.float64) + float(m["offset"])) * (float(m["range"]) / float(m["digitisation"]))

    def read_attrs(self):
        return dict(self._read_group().attrs)

    def stored_basecall(self, section="template"):
        """(name, sequence, quality) of the 1D basecall ONT's software stored in the file, or None."""
        analyses = self.h5.root["Analyses"] if "Analyses" in self.h5.root else None
        if analyses is None:
            return None
        for grp in sorted(k for k in analyses.keys() if k.startswith("Basecall_1D")):
            p = "Analyses/%s/BaseCalled_%s/Fastq" % (grp, section)
            if p in self.h5.root:
                fq = self.h5.read(self.h5.root[p])
                if isinstance(fq, bytes):
                    fq = fq.decode("utf-8", "replace")
                lines = fq.strip().split("\n")
                if len(lines) >= 4:
                    return lines[0][1:], lines[1], lines[3]
        return None
