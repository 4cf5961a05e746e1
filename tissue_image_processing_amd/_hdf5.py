"""A self-contained reader for the HDF5 files Keras writes its weights to (pl.py:76-88: model.load_weights(path) with the
`.h5` checkpoint gui.py:38-39 names).  No h5py / TensorFlow in this image, so the container format is read here, from the
published HDF5 File Format Specification (version 3.0), as far as such files use it:

  * superblock versions 0 / 1 (h5py's default `libver='earliest'`, what Keras writes) and 2 / 3 (`libver='latest'`);
  * groups: symbol-table groups (object header v1, B-tree v1 'TREE' of 'SNOD' nodes, names in a local 'HEAP') and new-style
    groups (object header v2 'OHDR' / 'OCHK', compact link messages, or dense links: fractal heap 'FRHP' / 'FHDB' / 'FHIB'
    indexed by a B-tree v2 'BTHD' / 'BTLF' / 'BTIN');
  * attributes: compact attribute messages (versions 1-3) and dense attribute storage (same fractal heap + B-tree v2 walk);
    fixed-length and variable-length (global heap 'GCOL') strings, integers, floats -- Keras' `layer_names`, `weight_names`,
    `backend`, `keras_version`;
  * datasets: little- or big-endian integers and floats, compact / contiguous / chunked (B-tree v1 chunk index; layout v4:
    single chunk, implicit, fixed array without paging) layouts, the deflate, shuffle and fletcher32 filters.

It is a reader only and only of what is listed; anything else raises Hdf5Error naming the feature.  tests/test_hdf5_reader.py
checks it against files written by the real HDF5 library (h5py of the golden interpreter: tools/make_h5_fixtures.py).
"""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(OSError):
    pass


class _Buf:
    """little-endian cursor over bytes"""

    def __init__(self, data, pos=0):
        self.d, self.p = data, pos

    def u(self, n):
        v = int.from_bytes(self.d[self.p:self.p + n], "little")
        if self.p + n > len(self.d):
            raise Hdf5Error("truncated HDF5 structure")
        self.p += n
        return v

    def raw(self, n):
        if self.p + n > len(self.d):
            raise Hdf5Error("truncated HDF5 structure")
        v = self.d[self.p:self.p + n]
        self.p += n
        return v

    def skip(self, n):
        self.p += n

    def align(self, n, base=0):
        r = (self.p - base) % n
        if r:
            self.p += n - r


class _Datatype:
    def __init__(self, cls, size, np_dtype=None, vlen_string=False, vlen_base=None, strpad=0):
        self.cls, self.size, self.np_dtype, self.vlen_string, self.vlen_base, self.strpad = cls, size, np_dtype, vlen_string, vlen_base, strpad


def _parse_datatype(b):
    """datatype message at cursor b -> _Datatype (advances b past the properties it understands)"""
    cv = b.u(1)
    cls, ver = cv & 15, cv >> 4
    bits = b.u(3)
    size = b.u(4)
    if cls == 0:                                   # fixed-point
        b.skip(4)
        order = ">" if bits & 1 else "<"
        kind = "i" if bits & 8 else "u"
        if size not in (1, 2, 4, 8):
            raise Hdf5Error("HDF5 integer of %d bytes is not supported" % size)
        return _Datatype(cls, size, np.dtype(order + kind + str(size)))
    if cls == 1:                                   # floating point
        b.skip(12)
        if bits & 0x40:
            raise Hdf5Error("VAX-order HDF5 floats are not supported")
        order = ">" if bits & 1 else "<"
        if size not in (2, 4, 8):
            raise Hdf5Error("HDF5 float of %d bytes is not supported" % size)
        return _Datatype(cls, size, np.dtype(order + "f" + str(size)))
    if cls == 3:                                   # fixed-length string
        return _Datatype(cls, size, np.dtype("S%d" % size), strpad=bits & 15)
    if cls == 9:                                   # variable length: sequence or string
        is_string = (bits & 15) == 1
        base = _parse_datatype(b)
        return _Datatype(cls, size, None, vlen_string=is_string, vlen_base=base)
    if cls == 8:                                   # enumeration (h5py stores numpy bool as an enum of int8)
        base = _parse_datatype(b)
        return _Datatype(cls, size, base.np_dtype)
    raise Hdf5Error("HDF5 datatype class %d (version %d) is not supported" % (cls, ver))


def _parse_dataspace(b):
    ver = b.u(1)
    rank = b.u(1)
    flags = b.u(1)
    if ver == 1:
        b.skip(5)
    elif ver == 2:
        typ = b.u(1)
        if typ == 2:
            return None                            # null dataspace
    else:
        raise Hdf5Error("HDF5 dataspace message version %d is not supported" % ver)
    return tuple(b.u(8) for _ in range(rank)), flags   # (sizes of lengths are 8 in every file this reader accepts)


class Hdf5Object:
    """A group or a dataset: `attrs` (dict), and for groups `keys()` / `[name]`, for datasets `shape`, `dtype`, `read()`."""

    def __init__(self, f, addr, name):
        self._f, self._addr, self.name = f, addr, name
        self._msgs = f._object_messages(addr)
        self._links = None
        self._attrs = None

    # -- attributes -----------------------------------------------------------------------------------------------------
    @property
    def attrs(self):
        if self._attrs is None:
            out = {}
            for typ, data, _ in self._msgs:
                if typ == 0x000C:
                    k, v = self._f._parse_attribute(data)
                    out[k] = v
                elif typ == 0x0015:                # attribute info: dense storage
                    b = _Buf(data)
                    ver, flags = b.u(1), b.u(1)
                    if flags & 1:
                        b.skip(2)
                    heap, btree = b.u(8), b.u(8)
                    if heap != UNDEF:
                        for blob in self._f._dense_objects(heap, btree, 8):
                            k, v = self._f._parse_attribute(blob)
                            out[k] = v
            self._attrs = out
        return self._attrs

    # -- group ----------------------------------------------------------------------------------------------------------
    def _load_links(self):
        if self._links is not None:
            return
        links = {}
        for typ, data, _ in self._msgs:
            if typ == 0x0011:                      # symbol table: B-tree v1 + local heap
                b = _Buf(data)
                btree, heap = b.u(8), b.u(8)
                links.update(self._f._symbol_table_links(btree, heap))
            elif typ == 0x0006:                    # link message
                k, a = self._f._parse_link(data)
                if a is not None:
                    links[k] = a
            elif typ == 0x0002:                    # link info: dense links
                b = _Buf(data)
                ver, flags = b.u(1), b.u(1)
                if flags & 1:
                    b.skip(8)
                heap, btree = b.u(8), b.u(8)
                if heap != UNDEF:
                    for blob in self._f._dense_objects(heap, btree, 5):
                        k, a = self._f._parse_link(blob)
                        if a is not None:
                            links[k] = a
        self._links = links

    @property
    def is_dataset(self):
        return any(t == 0x0008 for t, _, _ in self._msgs)

    def keys(self):
        self._load_links()
        return list(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        obj = self
        for part in [p for p in path.split("/") if p]:
            obj._load_links()
            if part not in obj._links:
                raise KeyError("%s: no object '%s' in HDF5 group '%s'" % (self._f.path, part, obj.name))
            obj = Hdf5Object(self._f, obj._links[part], (obj.name.rstrip("/") + "/" + part))
        return obj

    # -- dataset --------------------------------------------------------------------------------------------------------
    def _dataset_info(self):
        dt = sp = layout = None
        filters = []
        for typ, data, _ in self._msgs:
            if typ == 0x0003:
                dt = _parse_datatype(_Buf(data))
            elif typ == 0x0001:
                sp = _parse_dataspace(_Buf(data))
            elif typ == 0x0008:
                layout = data
            elif typ == 0x000B:
                filters = self._f._parse_filters(data)
        if dt is None or layout is None:
            raise Hdf5Error("%s: '%s' is not a dataset" % (self._f.path, self.name))
        shape = () if sp is None else sp[0]
        return dt, shape, layout, filters

    @property
    def shape(self):
        return self._dataset_info()[1]

    @property
    def dtype(self):
        return self._dataset_info()[0].np_dtype

    def read(self):
        dt, shape, layout, filters = self._dataset_info()
        if dt.np_dtype is None:
            raise Hdf5Error("%s: dataset '%s' has a variable-length type" % (self._f.path, self.name))
        raw = self._f._read_layout(layout, shape, dt.size, filters)
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        arr = np.frombuffer(raw, dtype=dt.np_dtype, count=n).reshape(shape)
        return arr.astype(dt.np_dtype.newbyteorder("=")) if dt.np_dtype.byteorder == ">" else arr.copy()


class Hdf5File(Hdf5Object):
    def __init__(self, path):
        self.path = path
        try:
            with open(path, "rb") as fh:
                self._d = fh.read()
        except OSError as e:
            raise OSError("Unable to open file (unable to open file: name = '%s', %s)" % (path, e.strerror or e))
        d = self._d
        off = 0
        while True:                                # the superblock sits at 0, 512, 1024, ... (user block in front)
            if d[off:off + 8] == SIGNATURE:
                break
            off = 512 if off == 0 else off * 2
            if off + 8 > len(d):
                raise Hdf5Error("Unable to open file (file signature not found): '%s' is not an HDF5 file" % path)
        b = _Buf(d, off + 8)
        ver = b.u(1)
        if ver in (0, 1):
            b.skip(4)
            so, sl = b.u(1), b.u(1)
            b.skip(1 + 2 + 2 + 4)
            if ver == 1:
                b.skip(4)
            self._check_sizes(so, sl)
            self._base = b.u(8)
            b.skip(8 * 3)
            b.skip(8)                              # root symbol table entry: link name offset
            root = b.u(8)
        elif ver in (2, 3):
            so, sl = b.u(1), b.u(1)
            b.skip(1)
            self._check_sizes(so, sl)
            self._base = b.u(8)
            b.skip(16)
            root = b.u(8)
        else:
            raise Hdf5Error("HDF5 superblock version %d is not supported" % ver)
        if self._base not in (0, off):
            raise Hdf5Error("HDF5 base address %d is not supported" % self._base)
        self._base = off if self._base == off else 0
        self._heaps = {}
        Hdf5Object.__init__(self, self, root, "/")

    @staticmethod
    def _check_sizes(so, sl):
        if so != 8 or sl != 8:
            raise Hdf5Error("HDF5 files with %d-byte offsets / %d-byte lengths are not supported (8 / 8 only)" % (so, sl))

    def _at(self, addr):
        if addr == UNDEF or addr + self._base >= len(self._d):
            raise Hdf5Error("%s: address %#x lies outside the file" % (self.path, addr))
        return _Buf(self._d, addr + self._base)

    # -- object headers -------------------------------------------------------------------------------------------------
    def _object_messages(self, addr):
        """[(type, data bytes, flags)] of the object header at addr, continuation blocks followed"""
        b = self._at(addr)
        msgs = []
        if b.d[b.p:b.p + 4] == b"OHDR":
            b.skip(4)
            if b.u(1) != 2:
                raise Hdf5Error("unknown object header version")
            flags = b.u(1)
            if flags & 0x20:
                b.skip(16)
            if flags & 0x10:
                b.skip(4)
            size = b.u(1 << (flags & 3))
            blocks = [(b.p, size)]
            order = bool(flags & 0x04)
            while blocks:
                start, length = blocks.pop(0)
                c = _Buf(self._d, start)
                end = start + length
                while c.p + 4 <= end:
                    typ, msize, mflags = c.u(1), c.u(2), c.u(1)
                    if order:
                        c.skip(2)
                    data = c.raw(msize)
                    if typ == 0x10:
                        cb = _Buf(data)
                        caddr, clen = cb.u(8), cb.u(8)
                        cc = self._at(caddr)
                        if cc.raw(4) != b"OCHK":
                            raise Hdf5Error("object header continuation without its signature")
                        blocks.append((cc.p, clen - 8))      # (signature and checksum are not messages)
                    elif typ != 0:
                        msgs.append((typ, data, mflags))
            return msgs
        ver = b.u(1)
        if ver != 1:
            raise Hdf5Error("%s: no object header at %#x" % (self.path, addr))
        b.skip(1)
        nmsg = b.u(2)
        b.skip(4)
        size = b.u(4)
        b.skip(4)                                  # (the 12-byte prefix is padded to 16)
        blocks = [(b.p, size)]
        while blocks and len(msgs) < nmsg + 64:
            start, length = blocks.pop(0)
            c = _Buf(self._d, start)
            end = start + length
            while c.p + 8 <= end:
                typ, msize, mflags = c.u(2), c.u(2), c.u(1)
                c.skip(3)
                data = c.raw(msize)
                if typ == 0x10:
                    cb = _Buf(data)
                    caddr, clen = cb.u(8), cb.u(8)
                    blocks.append((self._at(caddr).p, clen))
                elif typ != 0:
                    msgs.append((typ, data, mflags))
        return msgs

    # -- old-style groups -----------------------------------------------------------------------------------------------
    def _symbol_table_links(self, btree, heap):
        hb = self._at(heap)
        if hb.raw(4) != b"HEAP":
            raise Hdf5Error("local heap signature missing")
        hb.skip(4)
        hb.skip(16)
        hdata = self._at(hb.u(8)).p
        out = {}

        def name_at(o):
            e = self._d.index(b"\0", hdata + o)
            return self._d[hdata + o:e].decode("utf-8")

        def node(addr):
            b = self._at(addr)
            sig = b.raw(4)
            if sig == b"SNOD":
                b.skip(2)
                n = b.u(2)
                for _ in range(n):
                    no, oh = b.u(8), b.u(8)
                    b.skip(24)
                    out[name_at(no)] = oh
                return
            if sig != b"TREE":
                raise Hdf5Error("group B-tree node signature missing")
            ntype, level, used = b.u(1), b.u(1), b.u(2)
            if ntype != 0:
                raise Hdf5Error("chunk B-tree where a group B-tree was expected")
            b.skip(16)
            for _ in range(used):
                b.skip(8)                          # key
                node(b.u(8))

        node(btree)
        return out

    # -- new-style groups / dense storage --------------------------------------------------------------------------------
    def _parse_link(self, data):
        b = _Buf(data)
        if b.u(1) != 1:
            raise Hdf5Error("unknown link message version")
        flags = b.u(1)
        ltype = b.u(1) if flags & 0x08 else 0
        if flags & 0x04:
            b.skip(8)
        if flags & 0x10:
            b.skip(1)
        n = b.u(1 << (flags & 3))
        name = b.raw(n).decode("utf-8")
        if ltype != 0:
            return name, None                      # soft / external links: not followed
        return name, b.u(8)

    def _fractal_heap(self, addr):
        if addr in self._heaps:
            return self._heaps[addr]
        b = self._at(addr)
        if b.raw(4) != b"FRHP":
            raise Hdf5Error("fractal heap signature missing")
        b.skip(1)
        h = {"id_len": b.u(2), "filt_len": b.u(2), "flags": b.u(1), "max_managed": b.u(4)}
        b.skip(8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8)
        h["width"], h["start"], h["max_direct"], h["max_bits"] = b.u(2), b.u(8), b.u(8), b.u(2)
        b.skip(2)
        h["root"], h["rows"] = b.u(8), b.u(2)
        if h["filt_len"]:
            raise Hdf5Error("filtered fractal heaps are not supported")
        h["off_bytes"] = (h["max_bits"] + 7) // 8
        lim = min(h["max_direct"], h["max_managed"])
        h["len_bytes"] = (max(lim.bit_length(), 1) + 7) // 8
        # direct blocks: [(heap offset, size, file position of the block start)]
        blocks = []

        def row_size(r):
            return h["start"] if r < 2 else h["start"] << (r - 1)

        def direct(a, size):
            d = self._at(a)
            if d.raw(4) != b"FHDB":
                raise Hdf5Error("fractal heap direct block signature missing")
            d.skip(1 + 8)
            blocks.append((d.u(h["off_bytes"]), size, a + self._base))

        def indirect(a, nrows):
            d = self._at(a)
            if d.raw(4) != b"FHIB":
                raise Hdf5Error("fractal heap indirect block signature missing")
            d.skip(1 + 8)
            d.skip(h["off_bytes"])
            max_direct_rows = 0
            while row_size(max_direct_rows) <= h["max_direct"]:
                max_direct_rows += 1
            for r in range(nrows):
                for _ in range(h["width"]):
                    child = d.u(8)
                    if child == UNDEF:
                        continue
                    if r < max_direct_rows:
                        direct(child, row_size(r))
                    else:
                        sub_rows = (row_size(r) // h["start"] // h["width"]).bit_length()      # rows of an indirect block covering row_size(r)
                        indirect(child, sub_rows)

        if h["root"] != UNDEF:
            if h["rows"] == 0:
                direct(h["root"], h["start"])
            else:
                indirect(h["root"], h["rows"])
        h["blocks"] = blocks
        self._heaps[addr] = h
        return h

    def _heap_object(self, h, hid):
        kind = (hid[0] >> 4) & 3
        if kind == 2:                              # tiny: the data sits in the ID
            n = (hid[0] & 15) + 1
            return bytes(hid[1:1 + n])
        if kind != 0:
            raise Hdf5Error("huge fractal-heap objects are not supported")
        off = int.from_bytes(hid[1:1 + h["off_bytes"]], "little")
        length = int.from_bytes(hid[1 + h["off_bytes"]:1 + h["off_bytes"] + h["len_bytes"]], "little")
        for boff, size, pos in h["blocks"]:
            if boff <= off < boff + size:
                return self._d[pos + off - boff:pos + off - boff + length]
        raise Hdf5Error("fractal heap object outside every direct block")

    def _dense_objects(self, heap_addr, btree_addr, btype):
        """the heap objects a B-tree v2 name index (type 5: links, 8: attributes) points at"""
        h = self._fractal_heap(heap_addr)
        b = self._at(btree_addr)
        if b.raw(4) != b"BTHD":
            raise Hdf5Error("B-tree v2 header signature missing")
        b.skip(1)
        typ = b.u(1)
        node_size, rec_size, depth = b.u(4), b.u(2), b.u(2)
        b.skip(2)
        root, nroot = b.u(8), b.u(2)
        if typ != btype:
            raise Hdf5Error("B-tree v2 of type %d where type %d was expected" % (typ, btype))
        out = []
        if root == UNDEF or nroot == 0:
            return out
        id_at = 4 if btype == 5 else 0             # type 5 record: hash(4) + heap ID(7); type 8: heap ID(8) + flags(1) + order(4) + hash(4)
        id_len = 7 if btype == 5 else 8

        def records(c, n):
            for _ in range(n):
                rec = c.raw(rec_size)
                out.append(self._heap_object(h, rec[id_at:id_at + id_len]))

        max_leaf = (node_size - 10) // rec_size
        leaf_bytes = (max(max_leaf.bit_length(), 1) + 7) // 8

        def node(addr, n, d):
            c = self._at(addr)
            sig = c.raw(4)
            c.skip(2)
            if d == 0:
                if sig != b"BTLF":
                    raise Hdf5Error("B-tree v2 leaf signature missing")
                records(c, n)
                return
            if sig != b"BTIN":
                raise Hdf5Error("B-tree v2 internal node signature missing")
            if d > 1:
                raise Hdf5Error("B-tree v2 deeper than two levels is not supported")
            records(c, n)
            for _ in range(n + 1):
                child, cn = c.u(8), c.u(leaf_bytes)
                node(child, cn, d - 1)

        node(root, nroot, depth)
        return out

    # -- attributes -----------------------------------------------------------------------------------------------------
    def _parse_attribute(self, data):
        b = _Buf(data)
        ver = b.u(1)
        b.skip(1)
        nlen, dlen, slen = b.u(2), b.u(2), b.u(2)
        if ver == 3:
            b.skip(1)
        elif ver not in (1, 2):
            raise Hdf5Error("attribute message version %d is not supported" % ver)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = b.raw(nlen).split(b"\0")[0].decode("utf-8")
        b.skip(pad(nlen) - nlen)
        dt = _parse_datatype(_Buf(b.raw(dlen)))
        b.skip(pad(dlen) - dlen)
        sp = _parse_dataspace(_Buf(b.raw(slen)))
        b.skip(pad(slen) - slen)
        shape = () if sp is None else sp[0]
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if sp is None:
            return name, None
        if dt.cls == 9:
            vals = []
            for _ in range(n):
                ln, coll, idx = b.u(4), b.u(8), b.u(4)
                blob = self._global_heap_object(coll, idx) if ln else b""
                if dt.vlen_string:
                    vals.append(blob[:ln].decode("utf-8"))
                else:
                    vals.append(np.frombuffer(blob, dtype=dt.vlen_base.np_dtype, count=ln).copy())
            if not shape:
                return name, vals[0]
            arr = np.empty(n, dtype=object)
            arr[:] = vals
            return name, arr.reshape(shape)
        arr = np.frombuffer(b.raw(n * dt.size), dtype=dt.np_dtype, count=n).reshape(shape).copy()
        if dt.cls == 3 and dt.strpad == 0:         # null-terminated: numpy's S dtype strips trailing nulls itself
            pass
        return name, (arr[()] if not shape else arr)

    def _global_heap_object(self, coll, idx):
        b = self._at(coll)
        if b.raw(4) != b"GCOL":
            raise Hdf5Error("global heap collection signature missing")
        b.skip(4)
        size = b.u(8)
        end = coll + self._base + size
        while b.p + 16 <= end:
            i = b.u(2)
            b.skip(6)
            n = b.u(8)
            if i == idx:
                return b.raw(n)
            if i == 0:
                break
            b.skip((n + 7) & ~7)
        raise Hdf5Error("global heap object %d not found" % idx)

    # -- dataset storage ------------------------------------------------------------------------------------------------
    @staticmethod
    def _parse_filters(data):
        b = _Buf(data)
        ver, n = b.u(1), b.u(1)
        if ver == 1:
            b.skip(6)
        elif ver != 2:
            raise Hdf5Error("filter pipeline version %d is not supported" % ver)
        out = []
        for _ in range(n):
            fid = b.u(2)
            nlen = b.u(2) if (ver == 1 or fid >= 256) else 0
            b.skip(2)
            ncl = b.u(2)
            if nlen:
                b.skip((nlen + 7) & ~7 if ver == 1 else nlen)
            cl = [b.u(4) for _ in range(ncl)]
            if ver == 1 and ncl % 2:
                b.skip(4)
            out.append((fid, cl))
        return out

    @staticmethod
    def _unfilter(raw, filters, mask, elem):
        for k in range(len(filters) - 1, -1, -1):
            if mask & (1 << k):
                continue
            fid, cl = filters[k]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cl[0] if cl else elem
                n = len(raw) // es
                a = np.frombuffer(raw[:n * es], dtype=np.uint8).reshape(es, n)
                raw = a.T.tobytes() + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise Hdf5Error("HDF5 filter %d is not supported (deflate, shuffle and fletcher32 are)" % fid)
        return raw

    def _read_layout(self, layout, shape, elem, filters):
        b = _Buf(layout)
        ver = b.u(1)
        if ver not in (3, 4):
            raise Hdf5Error("data layout message version %d is not supported" % ver)
        cls = b.u(1)
        total = (int(np.prod(shape, dtype=np.int64)) if shape else 1) * elem
        if cls == 0:
            return b.raw(b.u(2))
        if cls == 1:
            addr, size = b.u(8), b.u(8)
            if addr == UNDEF:
                return bytes(total)                # never written: fill value 0
            s = addr + self._base
            return self._d[s:s + total]
        if cls != 2:
            raise Hdf5Error("data layout class %d is not supported" % cls)
        rank = len(shape)
        out = np.zeros(shape if shape else (1,), dtype=np.uint8 if elem == 1 else "V%d" % elem)

        def place(raw, offs, cdims):
            chunk = np.frombuffer(raw, dtype=out.dtype, count=int(np.prod(cdims))).reshape(cdims)
            sel_o = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
            sel_c = tuple(slice(0, min(o + c, s) - o) for o, c, s in zip(offs, cdims, shape))
            out[sel_o] = chunk[sel_c]

        if ver == 3:
            nd = b.u(1)
            btree = b.u(8)
            cdims = tuple(b.u(4) for _ in range(nd))[:-1]
            if len(cdims) != rank:
                raise Hdf5Error("chunk rank does not match the dataspace")

            def node(addr):
                c = self._at(addr)
                if c.raw(4) != b"TREE":
                    raise Hdf5Error("chunk B-tree signature missing")
                ntype, level, used = c.u(1), c.u(1), c.u(2)
                c.skip(16)
                for _ in range(used):
                    csize, mask = c.u(4), c.u(4)
                    offs = tuple(c.u(8) for _ in range(rank + 1))[:-1]
                    child = c.u(8)
                    if level > 0:
                        node(child)
                    else:
                        s = child + self._base
                        place(self._unfilter(self._d[s:s + csize], filters, mask, elem), offs, cdims)

            if btree != UNDEF:
                node(btree)
            return out.tobytes()
        # layout version 4
        flags, nd, enc = b.u(1), b.u(1), b.u(1)
        cdims = tuple(b.u(enc) for _ in range(nd))[:-1]
        if len(cdims) != rank:
            raise Hdf5Error("chunk rank does not match the dataspace")
        index = b.u(1)
        csize = int(np.prod(cdims)) * elem
        grid = [(-(-s // c)) for s, c in zip(shape, cdims)]
        nchunks = int(np.prod(grid))

        def offs_of(i):
            o = []
            for g, c in zip(reversed(grid), reversed(cdims)):
                o.append((i % g) * c)
                i //= g
            return tuple(reversed(o))

        if index == 1:                             # single chunk
            if flags & 2:
                fsize, mask = b.u(8), b.u(4)
            else:
                fsize, mask = csize, 0
            addr = b.u(8)
            if addr != UNDEF:
                s = addr + self._base
                place(self._unfilter(self._d[s:s + fsize], filters if flags & 2 else [], mask, elem), (0,) * rank, cdims)
            return out.tobytes()
        if index == 2:                             # implicit: chunks back to back, no filters
            addr = b.u(8)
            if addr != UNDEF:
                for i in range(nchunks):
                    s = addr + self._base + i * csize
                    place(self._d[s:s + csize], offs_of(i), cdims)
            return out.tobytes()
        if index == 3:                             # fixed array
            page_bits = b.u(1)
            c = self._at(b.u(8))
            if c.raw(4) != b"FAHD":
                raise Hdf5Error("fixed array header signature missing")
            c.skip(1)
            client, esize, pbits = c.u(1), c.u(1), c.u(1)
            nelem, dblk = c.u(8), c.u(8)
            if dblk == UNDEF:
                return out.tobytes()
            c = self._at(dblk)
            if c.raw(4) != b"FADB":
                raise Hdf5Error("fixed array data block signature missing")
            c.skip(1 + 1 + 8)
            if nelem > (1 << pbits):
                raise Hdf5Error("paged fixed-array chunk indexes are not supported")
            for i in range(nelem):
                if client == 0:
                    addr, fsize, mask = c.u(8), csize, 0
                else:
                    addr = c.u(8)
                    fsize = c.u(esize - 8 - 4)
                    mask = c.u(4)
                if addr != UNDEF:
                    s = addr + self._base
                    place(self._unfilter(self._d[s:s + fsize], filters if client else [], mask, elem), offs_of(i), cdims)
            return out.tobytes()
        raise Hdf5Error("chunk index type %d (extensible array / B-tree v2) is not supported" % index)


def load_keras_weights_h5(path):
    """The weights of a Keras `.h5` checkpoint -- model.save_weights(path) or model.save(path) in the HDF5 format -- as
    [(layer name, [(weight name, array), ...])] in the file's layer order, layers without weights included (empty lists).

    Layout (keras/saving/hdf5_format.py, `save_weights_to_hdf5_group`): the root group -- or its `model_weights` group in a
    whole-model file -- carries the attribute `layer_names`; every layer is a group of that name whose attribute
    `weight_names` lists its datasets (`<layer>/kernel:0`, ...), in model.get_weights() order.  Attributes too large for one
    object-header message are split into `layer_names0`, `layer_names1`, ... (`save_attributes_to_hdf5_group`)."""
    f = Hdf5File(path)
    g = f
    if "layer_names" not in f.attrs and "layer_names0" not in f.attrs and "model_weights" in f:
        g = f["model_weights"]

    def chunked_attr(obj, name):
        if name in obj.attrs:
            vals = list(np.atleast_1d(obj.attrs[name]))
        else:
            vals, k = [], 0
            while "%s%d" % (name, k) in obj.attrs:
                vals.extend(np.atleast_1d(obj.attrs["%s%d" % (name, k)]))
                k += 1
            if k == 0:
                raise Hdf5Error("%s: no '%s' attribute on '%s' -- not a Keras weights file" % (path, name, obj.name))
        return [v.decode("utf-8") if isinstance(v, (bytes, np.bytes_)) else str(v) for v in vals]

    layers = []
    for lname in chunked_attr(g, "layer_names"):
        lg = g[lname]
        layers.append((lname, [(wn, lg[wn].read()) for wn in chunked_attr(lg, "weight_names")]))
    return layers
