"""Read-only HDF5 for the NetCDF-4 files the `downscale` CLI is pointed at (/root/reference/src/downscaling/cli.py:22:
`xr.open_mfdataset` opens whatever netCDF4 / h5netcdf can; current Climate Data Store downloads are NetCDF-4, i.e. HDF5).

netCDF4 / h5py / libhdf5 are not part of the GPU image, so this is a from-the-format-specification reader in numpy +
zlib, restricted to what netCDF-C (and h5py / xarray's h5netcdf backend) writes for gridded data:

  * superblock versions 0-3; object headers version 1 and 2 with continuation blocks;
  * groups: symbol tables (B-tree v1 + local heap) and link messages, compact or dense (fractal heap + B-tree v2);
  * datasets: compact, contiguous and chunked layouts (B-tree v1 index; of the version-4 layouts the single-chunk,
    implicit, fixed-array and extensible-array indexes), fixed-point / floating-point / fixed-length string element
    types of either byte order, filters deflate, shuffle and fletcher32;
  * attributes, compact or dense: numeric, fixed- and variable-length strings, object references (DIMENSION_LIST).

Anything else (compound / enum / array element types, szip, B-tree-v2 chunk indexes, paged extensible-array blocks,
external links, virtual datasets) raises NotImplementedError naming the feature.  `tests/test_hdf5_cpu.py` pins it
against files written by libhdf5 1.10.6 itself (tests/golden/make_nc4_fixtures.py: what netCDF-C writes, the
libver-latest forms, a plain h5py-style file) and the arrays that were handed to the library.
"""
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(OSError):
    pass


def _u(buf, off, n):
    return int.from_bytes(buf[off:off + n], "little")


class _Type:
    """Decoded datatype message."""

    def __init__(self, cls, size, dtype=None, vlen_base=None, vlen_string=False, ref=False, strpad=0):
        self.cls, self.size, self.dtype, self.vlen_base, self.vlen_string, self.ref, self.strpad = \
            cls, size, dtype, vlen_base, vlen_string, ref, strpad


class Dataset:
    def __init__(self, f, name, addr, msgs):
        self.file, self.name, self.addr, self._msgs = f, name, addr, msgs
        self.shape = self.maxshape = ()
        self.type = None
        self.attrs = {}
        self._layout = None
        self._filters = []
        self._fill = None          # raw bytes of the fill value (messages 0x05 / 0x04), None when undefined

    @property
    def dtype(self):
        return self.type.dtype

    def __getitem__(self, key):
        return self.read()[key]

    def read(self):
        return self.file._read_dataset(self)


class Group:
    def __init__(self, f, name, addr):
        self.file, self.name, self.addr = f, name, addr
        self.attrs = {}
        self.members = {}          # name -> Group | Dataset

    def __getitem__(self, name):
        node = self
        for part in [p for p in name.split("/") if p]:
            node = node.members[part]
        return node

    def __contains__(self, name):
        return name in self.members

    def datasets(self):
        return {k: v for k, v in self.members.items() if isinstance(v, Dataset)}


class File:
    def __init__(self, path):
        # the file is MAPPED, not read: a multi-GB CDS download is touched page by page as datasets are decoded
        import mmap
        with open(str(path), "rb") as fh:
            try:
                self.buf = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            except (ValueError, OSError):              # empty file / no mmap on this filesystem
                self.buf = fh.read()
        self.path = str(path)
        base = 0
        while self.buf[base:base + 8] != SIGNATURE:          # (a user block: the superblock sits at 512, 1024, ...)
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self.buf):
                raise Hdf5Error(f"{path}: not an HDF5 file")
        b = self.buf
        ver = b[base + 8]
        if ver in (0, 1):
            self.O, self.L = b[base + 13], b[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = _u(b, p, self.O)
            p += 4 * self.O                                   # base, free-space info, end of file, driver info
            root_addr = _u(b, p + self.O, self.O)             # symbol-table entry: link-name offset, object header address
        elif ver in (2, 3):
            self.O, self.L = b[base + 9], b[base + 10]
            p = base + 12
            self.base = _u(b, p, self.O)
            root_addr = _u(b, p + 3 * self.O, self.O)
        else:
            raise NotImplementedError(f"{path}: HDF5 superblock version {ver}")
        if self.base not in (0, base):
            raise NotImplementedError(f"{path}: HDF5 base address {self.base}")
        self.base = base if self.base == base and base else self.base
        self._by_addr = {}
        self._gcol = {}
        self.root = self._load_object("/", root_addr)

    # ---- low level ---------------------------------------------------------------------------------
    def _a(self, off):
        """Offset-sized address at absolute position off (absolute file position returned; UNDEF -> None)."""
        v = _u(self.buf, off, self.O)
        return None if v == (1 << (8 * self.O)) - 1 else v + self.base

    def _messages(self, addr):
        """[(type, flags, payload bytes)] of the object header at addr (either version), continuations followed."""
        b = self.buf
        out = []
        if b[addr:addr + 4] == b"OHDR":
            if b[addr + 4] != 2:
                raise NotImplementedError(f"object header version {b[addr + 4]}")
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            nsz = 1 << (flags & 3)
            chunk = _u(b, p, nsz)
            p += nsz
            blocks = [(p, p + chunk)]
            corder = bool(flags & 0x04)
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 <= end:
                    mtype, msize, mflags = b[p], _u(b, p + 1, 2), b[p + 3]
                    p += 4 + (2 if corder else 0)
                    if p + msize > end:
                        break
                    data = b[p:p + msize]
                    p += msize
                    if mtype == 0x10:
                        ca, cl = self._a_bytes(data, 0), _u(data, self.O, self.L)
                        if b[ca:ca + 4] != b"OCHK":
                            raise Hdf5Error("object header continuation without OCHK signature")
                        blocks.append((ca + 4, ca + cl - 4))
                    elif mtype != 0:
                        out.append((mtype, mflags, data))
            return out
        if b[addr] != 1:
            raise Hdf5Error(f"{self.path}: no object header at {addr}")
        nmsg, hsize = _u(b, addr + 2, 2), _u(b, addr + 8, 4)
        blocks = [(addr + 16, addr + 16 + hsize)]
        while blocks and len(out) < nmsg + 64:
            p, end = blocks.pop(0)
            while p + 8 <= end:
                mtype, msize, mflags = _u(b, p, 2), _u(b, p + 2, 2), b[p + 4]
                p += 8
                data = b[p:p + msize]
                p += msize
                if mtype == 0x10:
                    blocks.append((self._a_bytes(data, 0), self._a_bytes(data, 0) + _u(data, self.O, self.L)))
                elif mtype != 0:
                    out.append((mtype, mflags, data))
        return out

    def _a_bytes(self, data, off):
        v = _u(data, off, self.O)
        return None if v == (1 << (8 * self.O)) - 1 else v + self.base

    # ---- datatype / dataspace ----------------------------------------------------------------------
    def _datatype(self, d, off=0):
        cv, bits, size = d[off], _u(d, off + 1, 3), _u(d, off + 4, 4)
        cls, ver = cv & 15, cv >> 4
        order = ">" if bits & 1 else "<"
        if cls == 0:
            kind = "i" if bits & 8 else "u"
            if size not in (1, 2, 4, 8):
                raise NotImplementedError(f"{size}-byte integers")
            return _Type(cls, size, np.dtype(f"{order}{kind}{size}"))
        if cls == 1:
            if size not in (2, 4, 8):
                raise NotImplementedError(f"{size}-byte floats")
            return _Type(cls, size, np.dtype(f"{order}f{size}"))
        if cls == 3:
            return _Type(cls, size, np.dtype(f"S{size}"), strpad=bits & 15)
        if cls == 7:
            if bits & 15:
                raise NotImplementedError("region references")
            return _Type(cls, size, np.dtype(f"<u{size}"), ref=True)
        if cls == 9:
            base = self._datatype(d, off + 8)
            return _Type(cls, size, None, vlen_base=base, vlen_string=(bits & 15) == 1)
        names = {2: "time", 4: "bitfield", 5: "opaque", 6: "compound", 8: "enum", 10: "array"}
        t = _Type(cls, size, None)
        t.unsupported = names.get(cls, f"class {cls}") + f" (version {ver})"
        return t

    def _dataspace(self, d):
        ver, rank, flags = d[0], d[1], d[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if d[3] == 2:
                return None, None         # null dataspace
            p = 4
        else:
            raise NotImplementedError(f"dataspace version {ver}")
        dims = tuple(_u(d, p + i * self.L, self.L) for i in range(rank))
        p += rank * self.L
        maxd = tuple(_u(d, p + i * self.L, self.L) for i in range(rank)) if flags & 1 else dims
        return dims, maxd

    # ---- heaps -------------------------------------------------------------------------------------
    def _global_heap_object(self, addr, index):
        col = self._gcol.get(addr)
        if col is None:
            b = self.buf
            if b[addr:addr + 4] != b"GCOL":
                raise Hdf5Error("global heap collection without GCOL signature")
            size = _u(b, addr + 8, self.L)
            p, end, col = addr + 8 + self.L, addr + size, {}
            while p + 8 + self.L <= end:
                idx, osz = _u(b, p, 2), _u(b, p + 8, self.L)
                if idx == 0:
                    break
                col[idx] = b[p + 8 + self.L:p + 8 + self.L + osz]
                p += 8 + self.L + (osz + 7) // 8 * 8
            self._gcol[addr] = col
        return col[index]

    def _vlen(self, raw, n, t):
        """n variable-length elements (length u32, collection address, index u32) -> list of str / arrays."""
        out, step = [], 4 + self.O + 4
        for i in range(n):
            ln, ga, gi = _u(raw, i * step, 4), self._a_bytes(raw, i * step + 4), _u(raw, i * step + 4 + self.O, 4)
            if ga is None or ln == 0 and gi == 0:
                out.append("" if t.vlen_string else np.zeros(0, t.vlen_base.dtype if t.vlen_base.dtype is not None else np.uint8))
                continue
            data = self._global_heap_object(ga, gi)
            if t.vlen_string:
                out.append(bytes(data[:ln]).split(b"\x00")[0].decode("utf-8", "replace"))
            else:
                out.append(np.frombuffer(data, t.vlen_base.dtype, ln).copy())
        return out

    class _Fractal:
        """Managed objects of a fractal heap (dense link / attribute storage)."""

        def __init__(self, f, addr):
            b, O, L = f.buf, f.O, f.L
            if b[addr:addr + 4] != b"FRHP":
                raise Hdf5Error("fractal heap without FRHP signature")
            self.f = f
            self.id_len, filt_len, self.flags = _u(b, addr + 5, 2), _u(b, addr + 7, 2), b[addr + 9]
            if filt_len:
                raise NotImplementedError("filtered fractal heaps")
            self.max_managed = _u(b, addr + 10, 4)
            p = addr + 14 + L + O + L + O + 4 * L + 4 * L          # huge id, huge btree, free space, fs manager, 4 managed counters, 4 huge/tiny counters
            self.width = _u(b, p, 2)
            self.start_size, self.max_direct = _u(b, p + 2, L), _u(b, p + 2 + L, L)
            self.heap_bits = _u(b, p + 2 + 2 * L, 2)
            p += 2 + 2 * L + 2 + 2
            self.root, self.root_rows = f._a(p), _u(b, p + O, 2)
            self.off_bytes = (self.heap_bits + 7) // 8
            max_dir_bits = max(self.max_direct.bit_length() - 1, 1)
            self.len_bytes = (min(max_dir_bits, max(self.max_managed.bit_length(), 1)) + 7) // 8
            self.blocks = []           # (heap offset, size, file address) of the direct blocks
            if self.root is not None:
                if self.root_rows == 0:
                    self.blocks.append((0, self.start_size, self.root))
                else:
                    self._indirect(self.root, self.root_rows)

        def _row_size(self, r):
            return self.start_size if r < 2 else self.start_size << (r - 1)

        def _indirect(self, addr, nrows):
            f, b = self.f, self.f.buf
            if b[addr:addr + 4] != b"FHIB":
                raise Hdf5Error("fractal heap indirect block without FHIB signature")
            off = _u(b, addr + 5 + f.O, self.off_bytes)
            p = addr + 5 + f.O + self.off_bytes
            max_direct_rows = (self.max_direct // self.start_size).bit_length() + 1
            for r in range(nrows):
                size = self._row_size(r)
                for _ in range(self.width):
                    child = f._a(p)
                    p += f.O
                    if r < max_direct_rows:
                        if child is not None:
                            self.blocks.append((off, size, child))
                    elif child is not None:
                        self._indirect(child, (size // self.start_size // self.width).bit_length())
                    off += size

        def get(self, heap_id):
            if (heap_id[0] >> 4) & 3:
                raise NotImplementedError("huge / tiny fractal-heap objects")
            off = _u(heap_id, 1, self.off_bytes)
            ln = _u(heap_id, 1 + self.off_bytes, self.len_bytes)
            for boff, bsize, baddr in self.blocks:
                if boff <= off < boff + bsize:
                    return self.f.buf[baddr + off - boff:baddr + off - boff + ln]
            raise Hdf5Error("fractal heap object outside every direct block")

    def _btree2_records(self, addr):
        """All records (raw bytes) of a version-2 B-tree."""
        b, O = self.buf, self.O
        if b[addr:addr + 4] != b"BTHD":
            raise Hdf5Error("B-tree v2 without BTHD signature")
        node_size, rec_size, depth = _u(b, addr + 6, 4), _u(b, addr + 10, 2), _u(b, addr + 12, 2)
        root, nroot = self._a(addr + 16), _u(b, addr + 16 + O, 2)
        nbytes = lambda v: max((int(v).bit_length() + 7) // 8, 1)
        max_n, cum = [(node_size - 10) // rec_size], [(node_size - 10) // rec_size]
        for d in range(1, depth + 1):
            ptr = O + nbytes(max_n[d - 1]) + (nbytes(cum[d - 1]) if d > 1 else 0)
            max_n.append((node_size - 10 - ptr) // (rec_size + ptr))
            cum.append((max_n[d] + 1) * cum[d - 1] + max_n[d])
        out = []

        def node(a, n, d):
            if a is None or n == 0:
                return
            sig = b"BTLF" if d == 0 else b"BTIN"
            if b[a:a + 4] != sig:
                raise Hdf5Error("B-tree v2 node signature")
            p = a + 6
            recs = [b[p + i * rec_size:p + (i + 1) * rec_size] for i in range(n)]
            p += n * rec_size
            if d == 0:
                out.extend(recs)
                return
            nb, tb = nbytes(max_n[d - 1]), (nbytes(cum[d - 1]) if d > 1 else 0)
            for i in range(n + 1):
                child, cn = self._a(p), _u(b, p + O, nb)
                p += O + nb + tb
                node(child, cn, d - 1)
                if i < n:
                    out.append(recs[i])

        node(root, nroot, depth)
        return out

    # ---- attributes --------------------------------------------------------------------------------
    def _decode_values(self, t, dims, raw):
        n = int(np.prod(dims)) if dims else 1
        if t.cls == 9:
            vals = self._vlen(raw, n, t)
            return vals[0] if not dims else (vals if t.vlen_string else vals)
        if getattr(t, "unsupported", None):
            return None
        arr = np.frombuffer(raw, t.dtype, n).copy()
        if t.cls == 3:
            strs = [bytes(x).split(b"\x00")[0].rstrip(b" " if t.strpad == 2 else b"").decode("utf-8", "replace") for x in arr]
            return strs[0] if not dims else strs
        arr = arr.astype(arr.dtype.newbyteorder("="))
        return arr.reshape(dims) if dims else arr.reshape(())[()]

    def _attribute(self, d):
        ver = d[0]
        nsz, tsz, ssz = _u(d, 2, 2), _u(d, 4, 2), _u(d, 6, 2)
        p = 8 + (1 if ver == 3 else 0)
        pad = (lambda v: (v + 7) // 8 * 8) if ver == 1 else (lambda v: v)
        name = bytes(d[p:p + nsz]).split(b"\x00")[0].decode("utf-8", "replace")
        p += pad(nsz)
        t = self._datatype(d, p)
        p += pad(tsz)
        dims, _ = self._dataspace(d[p:p + ssz])
        p += pad(ssz)
        if dims is None:
            return name, None
        return name, self._decode_values(t, dims, d[p:])

    def _attributes(self, msgs):
        attrs = {}
        for mtype, _, d in msgs:
            if mtype == 0x0C:
                k, v = self._attribute(d)
                attrs[k] = v
            elif mtype == 0x15:
                flags = d[1]
                p = 2 + (2 if flags & 1 else 0)
                heap, bt = self._a_bytes(d, p), self._a_bytes(d, p + self.O)
                if heap is None or bt is None:
                    continue
                fh = File._Fractal(self, heap)
                for rec in self._btree2_records(bt):
                    k, v = self._attribute(fh.get(rec[:fh.id_len]))
                    attrs[k] = v
        return attrs

    # ---- groups ------------------------------------------------------------------------------------
    def _link(self, d):
        flags = d[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = d[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nl = 1 << (flags & 3)
        n = _u(d, p, nl)
        p += nl
        name = bytes(d[p:p + n]).decode("utf-8", "replace")
        p += n
        return name, (self._a_bytes(d, p) if ltype == 0 else None)

    def _symbol_table(self, btree, heap):
        b, O, L = self.buf, self.O, self.L
        if b[heap:heap + 4] != b"HEAP":
            raise Hdf5Error("local heap without HEAP signature")
        data = self._a(heap + 8 + 2 * L)
        links = []

        def node(a):
            if b[a:a + 4] == b"SNOD":
                n = _u(b, a + 6, 2)
                p = a + 8
                for _ in range(n):
                    noff, oaddr = _u(b, p, O), self._a(p + O)
                    end = b.find(b"\x00", data + noff)           # (find: bytes and mmap both have it)
                    links.append((b[data + noff:end].decode("utf-8", "replace"), oaddr))
                    p += 2 * O + 24
                return
            if b[a:a + 4] != b"TREE":
                raise Hdf5Error("group B-tree node signature")
            n = _u(b, a + 6, 2)
            p = a + 8 + 2 * O + L
            for _ in range(n):
                node(self._a(p))
                p += O + L

        node(btree)
        return links

    def _load_object(self, name, addr):
        if addr in self._by_addr:
            return self._by_addr[addr]
        msgs = self._messages(addr)
        types = {m[0] for m in msgs}
        if 0x08 in types or (0x01 in types and 0x03 in types and 0x11 not in types and 0x02 not in types and 0x06 not in types):
            ds = Dataset(self, name, addr, msgs)
            self._by_addr[addr] = ds
            for mtype, _, d in msgs:
                if mtype == 0x01:
                    ds.shape, ds.maxshape = self._dataspace(d)
                elif mtype == 0x03:
                    ds.type = self._datatype(d)
                elif mtype == 0x08:
                    ds._layout = d
                elif mtype == 0x0B:
                    ds._filters = self._filter_pipeline(d)
                elif mtype == 0x05:
                    ds._fill = self._fill_value(d) or ds._fill
                elif mtype == 0x04 and ds._fill is None and len(d) >= 4:      # "fill value (old)": size, value
                    n_ = _u(d, 0, 4)
                    ds._fill = bytes(d[4:4 + n_]) if 0 < n_ <= len(d) - 4 else None
            ds.attrs = self._attributes(msgs)
            return ds
        g = Group(self, name, addr)
        self._by_addr[addr] = g
        links = []
        for mtype, _, d in msgs:
            if mtype == 0x11:
                links += self._symbol_table(self._a_bytes(d, 0), self._a_bytes(d, self.O))
            elif mtype == 0x06:
                links.append(self._link(d))
            elif mtype == 0x02:
                flags = d[1]
                p = 2 + (8 if flags & 1 else 0)
                heap, bt = self._a_bytes(d, p), self._a_bytes(d, p + self.O)
                if heap is not None and bt is not None:
                    fh = File._Fractal(self, heap)
                    for rec in self._btree2_records(bt):
                        links.append(self._link(fh.get(rec[4:4 + fh.id_len])))
        g.attrs = self._attributes(msgs)
        for lname, laddr in links:
            if laddr is not None:
                g.members[lname] = self._load_object((name.rstrip("/") + "/" + lname), laddr)
        return g

    # ---- dataset payload ---------------------------------------------------------------------------
    def _filter_pipeline(self, d):
        ver, n = d[0], d[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = _u(d, p, 2)
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = _u(d, p, 2)
                p += 2
            p += 2                                               # flags
            ncd = _u(d, p, 2)
            p += 2
            p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = [_u(d, p + 4 * i, 4) for i in range(ncd)]
            p += 4 * ncd
            if ver == 1 and ncd & 1:
                p += 4
            out.append((fid, cd))
        return out

    def _unfilter(self, data, filters, mask, itemsize):
        for i in range(len(filters) - 1, -1, -1):
            if mask & (1 << i):
                continue
            fid, cd = filters[i]
            if fid == 1:
                data = zlib.decompress(bytes(data))
            elif fid == 2:
                es = cd[0] if cd else itemsize
                n = len(data) // es
                body = np.frombuffer(data, np.uint8, n * es).reshape(es, n).T.tobytes()
                data = body + bytes(data[n * es:])
            elif fid == 3:
                data = data[:-4]
            elif fid == 4:
                raise NotImplementedError("szip-compressed chunks")
            else:
                raise NotImplementedError(f"HDF5 filter {fid}")
        return data

    def _chunks_btree1(self, addr, rank):
        """(chunk offsets, file address, stored size, filter mask) of every chunk under a version-1 chunk B-tree."""
        b, O = self.buf, self.O
        out = []
        ksz = 8 + 8 * (rank + 1)

        def node(a):
            if a is None:
                return
            if b[a:a + 4] != b"TREE" or b[a + 4] != 1:
                raise Hdf5Error("chunk B-tree node signature")
            level, n = b[a + 5], _u(b, a + 6, 2)
            p = a + 8 + 2 * O
            for _ in range(n):
                size, mask = _u(b, p, 4), _u(b, p + 4, 4)
                offs = tuple(_u(b, p + 8 + 8 * i, 8) for i in range(rank))
                child = self._a(p + ksz)
                p += ksz + O
                if level == 0:
                    out.append((offs, child, size, mask))
                else:
                    node(child)

        node(addr)
        return out

    @staticmethod
    def _fill_value(d):
        """Raw bytes of the value in a Fill Value message (0x05), or None when it is undefined / the library default (zeros).
        Versions 1 and 2: alloc time, write time, `defined` flag, then (v1 always, v2 when defined) size + value; version 3: one
        flags byte (bit 5: a value follows, bit 4: explicitly undefined)."""
        if len(d) < 2:
            return None
        ver = d[0]
        if ver in (1, 2):
            if len(d) < 4 or (ver == 2 and not d[3]):
                return None
            if len(d) < 8:
                return None
            n_ = _u(d, 4, 4)
            return bytes(d[8:8 + n_]) if 0 < n_ <= len(d) - 8 else None
        if ver == 3:
            if not (d[1] & 0x20) or len(d) < 6:
                return None
            n_ = _u(d, 2, 4)
            return bytes(d[6:6 + n_]) if 0 < n_ <= len(d) - 6 else None
        return None

    def _filled(self, ds, shape, native):
        """An array of the dataset's fill value: what libhdf5 returns for storage that was never written (a contiguous dataset
        without an address, a missing chunk).  The HDF5 message first; netCDF-4 repeats it as the `_FillValue` attribute.  For
        packed variables this is what decodes to NaN — zeros would decode to add_offset, a plausible value."""
        t = ds.type
        fv = None
        if ds._fill is not None and len(ds._fill) == t.size and t.dtype.kind in "iuf":
            fv = np.frombuffer(ds._fill, t.dtype, 1)[0]
        elif "_FillValue" in ds.attrs and t.dtype.kind in "iuf":
            try:
                fv = np.asarray(ds.attrs["_FillValue"]).reshape(-1)[0]
            except (IndexError, TypeError, ValueError):
                fv = None
        out = np.zeros(shape, native)
        if fv is not None:
            out[...] = np.asarray(fv).astype(native)
        return out

    def _read_dataset(self, ds):
        t = ds.type
        if t is None or ds._layout is None:
            raise Hdf5Error(f"{ds.name}: dataset without datatype / layout")
        if getattr(t, "unsupported", None) or t.cls == 9:
            raise NotImplementedError(f"{ds.name}: {getattr(t, 'unsupported', 'variable-length')} element type")
        shape = ds.shape or ()
        n = int(np.prod(shape)) if shape else 1
        d = ds._layout
        ver, cls = d[0], d[1]
        if ver not in (3, 4):
            raise NotImplementedError(f"{ds.name}: data layout version {ver}")
        native = t.dtype.newbyteorder("=") if t.dtype.kind in "iuf" else t.dtype
        if cls == 0:
            size = _u(d, 2, 2)
            return np.frombuffer(d[4:4 + size], t.dtype, n).astype(native).reshape(shape)
        if cls == 1:
            addr = self._a_bytes(d, 2)
            if addr is None:
                return self._filled(ds, shape, native)     # never written
            return np.frombuffer(self.buf, t.dtype, n, addr).astype(native).reshape(shape)
        if cls != 2:
            raise NotImplementedError(f"{ds.name}: data layout class {cls}")
        rank = len(shape)
        chunks = []
        if ver == 3:
            nd = d[2]
            bt = self._a_bytes(d, 3)
            cdims = tuple(_u(d, 3 + self.O + 4 * i, 4) for i in range(nd - 1))
            chunks = self._chunks_btree1(bt, rank)
        else:
            flags, nd, enc = d[2], d[3], d[4]
            cdims = tuple(_u(d, 5 + enc * i, enc) for i in range(nd - 1))
            p = 5 + enc * nd
            itype = d[p]
            p += 1
            csize = int(np.prod(cdims)) * t.size
            grid = [(-(-s // c)) for s, c in zip(shape, cdims)]
            if itype == 1:
                size, mask = csize, 0
                if flags & 2:
                    size, mask = _u(d, p, self.L), _u(d, p + self.L, 4)
                    p += self.L + 4
                chunks = [((0,) * rank, self._a_bytes(d, p), size, mask)]
            elif itype == 2:
                addr = self._a_bytes(d, p)
                for i, idx in enumerate(np.ndindex(*grid)):
                    chunks.append((tuple(a * c for a, c in zip(idx, cdims)), None if addr is None else addr + i * csize, csize, 0))
            elif itype == 3:
                chunks = self._chunks_fixed_array(self._a_bytes(d, p + 1), grid, cdims, csize)
            elif itype == 4:
                chunks = self._chunks_extensible_array(self._a_bytes(d, p + 5), grid, cdims, csize, ds.maxshape)
            else:
                raise NotImplementedError(f"{ds.name}: chunk index type {itype} (B-tree v2: written with libver='latest' on "
                                          f"several unlimited dimensions) — repack with `h5repack --low=0`")
        # (missing chunks keep the fill value; when every chunk is there the initial value is never seen)
        complete = len(chunks) > 0 and all(c[1] is not None for c in chunks) and \
            len(chunks) >= int(np.prod([-(-s_ // c_) for s_, c_ in zip(shape, cdims)])) if rank else True
        out = np.empty(shape, native) if complete else self._filled(ds, shape, native)
        for offs, addr, size, mask in chunks:
            if addr is None:
                continue
            raw = self._unfilter(self.buf[addr:addr + size], ds._filters, mask, t.size)
            block = np.frombuffer(raw, t.dtype, int(np.prod(cdims))).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
            if any(s.start >= s.stop for s in sl):
                continue
            out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def _chunks_fixed_array(self, hdr, grid, cdims, csize):
        b, O = self.buf, self.O
        if hdr is None:
            return []
        if b[hdr:hdr + 4] != b"FAHD":
            raise Hdf5Error("fixed array header signature")
        client, esize, page_bits = b[hdr + 5], b[hdr + 6], b[hdr + 7]
        nelm = _u(b, hdr + 8, self.L)
        db = self._a(hdr + 8 + self.L)
        if db is None:
            return []
        if b[db:db + 4] != b"FADB":
            raise Hdf5Error("fixed array data block signature")
        p = db + 6 + O
        page = 1 << page_bits
        paged = nelm > page
        npages = -(-nelm // page) if paged else 1
        bitmap = None
        if paged:
            nb = (npages + 7) // 8
            bitmap = b[p:p + nb]
            p += nb + 4                                       # + the data block's checksum (pages follow it)
        out = []
        idxs = list(np.ndindex(*grid))
        e = 0
        for pg in range(npages):
            cnt = min(page, nelm - pg * page) if paged else nelm
            if paged and not (bitmap[pg // 8] >> (7 - pg % 8)) & 1:
                e += cnt
                continue
            for _ in range(cnt):
                addr = self._a(p)
                size, mask = csize, 0
                if client == 1:
                    size, mask = _u(b, p + O, esize - O - 4), _u(b, p + esize - 4, 4)
                p += esize
                if e < len(idxs):
                    out.append((tuple(a * c for a, c in zip(idxs[e], cdims)), addr, size, mask))
                e += 1
            if paged:
                p += 4
        return out


    def _chunks_extensible_array(self, hdr, grid, cdims, csize, maxshape):
        """Chunk index of a dataset with ONE unlimited dimension written with libver >= 1.10: elements in chunk-index order — the
        first few in the index block, then data blocks of doubling sizes, the later ones reached through super blocks."""
        b, O, L = self.buf, self.O, self.L
        if hdr is None:
            return []
        if b[hdr:hdr + 4] != b"EAHD":
            raise Hdf5Error("extensible array header signature")
        client, esize, max_bits, n_idx, dmin, sup_min, page_bits = (b[hdr + 5 + i] for i in range(7))
        nelm = _u(b, hdr + 12 + 5 * L, L)
        iblk = self._a(hdr + 12 + 6 * L)
        if iblk is None:
            return []
        if b[iblk:iblk + 4] != b"EAIB":
            raise Hdf5Error("extensible array index block signature")
        off_bytes = (max_bits + 7) // 8
        log2 = lambda v: int(v).bit_length() - 1
        nsblks = 1 + max_bits - log2(dmin)
        iblock_nsblks = 2 * log2(sup_min)
        ndblk_addrs = 2 * (sup_min - 1)

        def element(p):
            addr = self._a(p)
            if client == 1:
                return addr, _u(b, p + O, esize - O - 4), _u(b, p + esize - 4, 4)
            return addr, csize, 0

        elems = []
        p = iblk + 6 + O
        for _ in range(n_idx):
            elems.append(element(p))
            p += esize
        dblk_addrs = [self._a(p + O * i) for i in range(ndblk_addrs)]
        p += O * ndblk_addrs
        sblk_addrs = [self._a(p + O * i) for i in range(nsblks - iblock_nsblks)]

        def data_block(addr, n):
            if addr is None:
                elems.extend([(None, 0, 0)] * n)
                return
            if b[addr:addr + 4] != b"EADB":
                raise Hdf5Error("extensible array data block signature")
            if n > (1 << page_bits):
                raise NotImplementedError("paged extensible-array data blocks (more than ~17,000 chunks along the unlimited dimension)")
            q = addr + 6 + O + off_bytes
            for _ in range(n):
                elems.append(element(q))
                q += esize

        next_d = 0
        for sb in range(nsblks):
            if len(elems) >= nelm:
                break
            ndb, nel = 1 << (sb // 2), (1 << ((sb + 1) // 2)) * dmin
            if sb < iblock_nsblks:
                for _ in range(ndb):
                    data_block(dblk_addrs[next_d], nel)
                    next_d += 1
                continue
            sa = sblk_addrs[sb - iblock_nsblks]
            if sa is None:
                elems.extend([(None, 0, 0)] * (ndb * nel))
                continue
            if b[sa:sa + 4] != b"EASB":
                raise Hdf5Error("extensible array super block signature")
            q = sa + 6 + O + off_bytes
            if nel > (1 << page_bits):
                raise NotImplementedError("paged extensible-array data blocks")
            for i in range(ndb):
                data_block(self._a(q + O * i), nel)
        # chunk index -> chunk coordinates: row-major with the unlimited dimension moved to the slowest position
        unl = [i for i, m in enumerate(maxshape) if m == (1 << (8 * L)) - 1]
        order = ([unl[0]] if unl else []) + [i for i in range(len(grid)) if not unl or i != unl[0]]
        out = []
        for e, idx in enumerate(np.ndindex(*[grid[i] for i in order])):
            if e >= len(elems):
                break
            addr, size, mask = elems[e]
            coord = [0] * len(grid)
            for j, i in enumerate(order):
                coord[i] = idx[j]
            out.append((tuple(a * c for a, c in zip(coord, cdims)), addr, size, mask))
        return out


def is_hdf5(path):
    with open(str(path), "rb") as f:
        return f.read(8) == SIGNATURE


# ---- the NetCDF-4 data model on top -----------------------------------------------------------------------
_NOT_A_VARIABLE = "This is a netCDF dimension but not a netCDF variable."


def read_netcdf4(path):
    """(coords, variables, attrs) of the root group of a NetCDF-4 file: coords {name: raw 1-D values + attrs}, variables
    {name: (dims, raw array, attrs)}.  Dimensions come from the dimension scales (DIMENSION_LIST object references, or the
    _Netcdf4Coordinates / _Netcdf4Dimid pair netCDF-C also writes); CF decoding is the caller's (io/netcdf.py)."""
    f = File(path)
    dsets = f.root.datasets()
    by_addr = {d.addr: k for k, d in dsets.items()}
    by_dimid = {int(np.asarray(d.attrs["_Netcdf4Dimid"]).reshape(-1)[0]): k for k, d in dsets.items() if "_Netcdf4Dimid" in d.attrs}
    scales = {k for k, d in dsets.items() if d.attrs.get("CLASS") == "DIMENSION_SCALE"}
    hidden = ("DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates", "_nc3_strict", "_NCProperties")
    out_coords, out_vars = {}, {}
    for name, d in dsets.items():
        attrs = {k: v for k, v in d.attrs.items() if k not in hidden}
        if name in scales and str(d.attrs.get("NAME", "")).startswith(_NOT_A_VARIABLE):
            continue                                           # a bare dimension: no values of its own
        dims = None
        dl = d.attrs.get("DIMENSION_LIST")
        if dl is not None and len(dl) == len(d.shape):
            try:
                dims = tuple(by_addr[int(np.asarray(r).reshape(-1)[0]) + f.base] for r in dl)
            except (KeyError, IndexError):
                dims = None
        if dims is None and "_Netcdf4Coordinates" in d.attrs:
            ids = [int(x) for x in np.asarray(d.attrs["_Netcdf4Coordinates"]).reshape(-1)]
            if all(i in by_dimid for i in ids):
                dims = tuple(by_dimid[i] for i in ids)
        if dims is None:
            dims = (name,) if name in scales and len(d.shape) == 1 else tuple(f"{name}_dim{i}" for i in range(len(d.shape)))
        if getattr(d.type, "unsupported", None) or d.type.cls == 9:
            continue                                           # (string / compound variables: not gridded data)
        data = d.read()
        if dims == (name,):
            out_coords[name] = (data, attrs)
        else:
            out_vars[name] = (dims, data, attrs)
    gattrs = {k: v for k, v in f.root.attrs.items() if k not in hidden}
    return out_coords, out_vars, gattrs
