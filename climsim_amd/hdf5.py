"""Dependency-free reader for the slice of HDF5 that ClimSim's files use (NetCDF-4 = HDF5).

The reference opens its raw timestep files and normalisation files with xarray / netCDF4 / h5py
(``climsim_utils/data_utils.py:619-640`` `get_xrdata`, ``:1029-1035`` `load_h5_file`;
``step2_retrain.py:187-190`` opens ``preprocessing/normalizations/{inputs,outputs}/*.nc``, which
are HDF5 files).  None of those packages is a dependency here, so this module reads the format
itself, from the published HDF5 file-format specification (version 3.0):

* superblock versions 0-3; object headers version 1 and 2 (with continuation blocks);
* groups: old style (symbol table: v1 B-tree + local heap + symbol nodes), new style compact
  (link messages) and dense (fractal heap; the link messages are read straight from the heap's
  direct blocks, the name-index B-tree is not needed to enumerate a group);
* datasets: fixed-point / floating-point / fixed-length string types of either byte order;
  compact, contiguous and chunked (v1 B-tree chunk index, layout message v3; layout v4 with a
  single chunk, implicit or fixed-array index) storage; deflate, shuffle and fletcher32 filters;
* attributes stored in the object header (`Hdf5File.attrs(name)`): numeric, fixed-length and variable-length string
  values (global heap) - what Keras' `.h5` checkpoints carry (`layer_names`, `weight_names`, `keras_version`,
  step2_retrain.py:253-261); attributes in dense storage are not read;
* references, variable-length and compound DATASET types are not read (not needed for `ds[var].values`); such datasets
  raise `Hdf5Unsupported` when accessed, the others stay readable.

`read_hdf5(path)` returns {name: ndarray} for every dataset below the root group (nested groups
joined with '/'); `Hdf5File` is the lazy form.
"""
from __future__ import annotations

import mmap
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class Hdf5Unsupported(NotImplementedError):
    pass


class _Dataset:
    __slots__ = ("shape", "dtype", "layout", "filters", "fill")

    def __init__(self):
        self.shape: Optional[Tuple[int, ...]] = None
        self.dtype = None
        self.layout = None
        self.filters: List[Tuple[int, Tuple[int, ...]]] = []
        self.fill = None


class Hdf5File:
    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        try:
            self.buf = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:                       # empty file
            self._f.close()
            raise ValueError(f"{path}: not an HDF5 file (empty)")
        self._objects: Dict[str, int] = {}       # dataset name -> object header address
        self._groups: Dict[str, int] = {}        # group path ('' = root) -> object header address
        self._read_superblock()
        self._walk(self.root_addr, "", set())

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if self.buf is not None:
            self.buf.close()
            self._f.close()
            self.buf = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _u(self, pos: int, size: int) -> int:
        return int.from_bytes(self.buf[pos:pos + size], "little")

    def _addr(self, pos: int) -> Optional[int]:
        v = self._u(pos, self.so)
        return None if v == (1 << (8 * self.so)) - 1 else v + self.base

    # ------------------------------------------------------------------ superblock
    def _read_superblock(self):
        b = self.buf
        pos = 0
        while True:                               # the superblock may sit at 0, 512, 1024, ...
            if b[pos:pos + 8] == SIGNATURE:
                break
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(b):
                raise ValueError(f"{self.path}: not an HDF5 file")
        ver = b[pos + 8]
        self.base = 0
        if ver in (0, 1):
            self.so, self.sl = b[pos + 13], b[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            self.base = self._u(p, self.so)
            p += 4 * self.so                      # base, free-space, end-of-file, driver-info
            # root group symbol table entry: link name offset, object header address, ...
            self.root_addr = self._u(p + self.so, self.so) + self.base
        elif ver in (2, 3):
            self.so, self.sl = b[pos + 9], b[pos + 10]
            p = pos + 12
            self.base = self._u(p, self.so)
            self.root_addr = self._u(p + 3 * self.so, self.so) + self.base
        else:
            raise Hdf5Unsupported(f"{self.path}: superblock version {ver}")

    # ------------------------------------------------------------------ object headers
    def _messages(self, addr: int) -> Iterator[Tuple[int, int, int]]:
        """(type, data position, data size) of every header message of the object at `addr`."""
        b = self.buf
        if b[addr:addr + 4] == b"OHDR":
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            csz = 1 << (flags & 3)
            chunk = self._u(p, csz)
            p += csz
            blocks = [(p, p + chunk)]
            extra = 2 if flags & 0x04 else 0
            i = 0
            while i < len(blocks):
                p, end = blocks[i]
                i += 1
                while p + 4 + extra <= end:
                    mtype, msize = b[p], self._u(p + 1, 2)
                    data = p + 4 + extra
                    if data + msize > end:
                        break
                    if mtype == 0x10:
                        caddr, clen = self._addr(data), self._u(data + self.so, self.sl)
                        if caddr is not None and b[caddr:caddr + 4] == b"OCHK":
                            blocks.append((caddr + 4, caddr + clen - 4))     # signature ... checksum
                    elif mtype != 0:
                        yield mtype, data, msize
                    p = data + msize
        else:
            if b[addr] != 1:
                raise Hdf5Unsupported(f"{self.path}: object header version {b[addr]} at {addr}")
            nmsg, hsize = self._u(addr + 2, 2), self._u(addr + 8, 4)
            blocks = [(addr + 16, addr + 16 + hsize)]
            i, seen = 0, 0
            while i < len(blocks) and seen < nmsg:
                p, end = blocks[i]
                i += 1
                while p + 8 <= end and seen < nmsg:
                    mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                    data = p + 8
                    seen += 1
                    if mtype == 0x10:
                        caddr, clen = self._addr(data), self._u(data + self.so, self.sl)
                        if caddr is not None:
                            blocks.append((caddr, caddr + clen))
                    elif mtype != 0:
                        yield mtype, data, msize
                    p = data + msize

    # ------------------------------------------------------------------ groups
    def _walk(self, addr: int, prefix: str, visiting: set):
        if addr in visiting:                      # hard-link cycles
            return
        visiting = visiting | {addr}
        msgs = list(self._messages(addr))
        types = {t for t, _, _ in msgs}
        if 0x08 in types or (0x01 in types and 0x03 in types):      # a dataset
            self._objects[prefix.rstrip("/")] = addr
            return
        self._groups[prefix.rstrip("/")] = addr
        for mtype, p, size in msgs:
            if mtype == 0x11:                     # symbol table: v1 B-tree + local heap
                btree, heap = self._addr(p), self._addr(p + self.so)
                if btree is not None and heap is not None:
                    for name, child in self._symbol_table(btree, heap):
                        self._walk(child, prefix + name + "/", visiting)
            elif mtype == 0x06:                   # link message (compact storage)
                link = self._link(p)
                if link:
                    self._walk(link[1], prefix + link[0] + "/", visiting)
            elif mtype == 0x02:                   # link info: dense storage in a fractal heap
                flags = self.buf[p + 1]
                q = p + 2 + (8 if flags & 1 else 0)
                heap = self._addr(q)
                if heap is not None:
                    for name, child in self._dense_links(heap):
                        self._walk(child, prefix + name + "/", visiting)

    def _link(self, p: int, end: Optional[int] = None) -> Optional[Tuple[str, int]]:
        """One link message -> (name, object header address); None for soft/external links."""
        b = self.buf
        if b[p] != 1:
            return None
        flags = b[p + 1]
        q = p + 2
        ltype = 0
        if flags & 0x08:
            ltype = b[q]
            q += 1
        if flags & 0x04:
            q += 8
        if flags & 0x10:
            q += 1
        nsz = 1 << (flags & 3)
        nlen = self._u(q, nsz)
        q += nsz
        name = bytes(b[q:q + nlen]).decode("utf-8", "replace")
        q += nlen
        if ltype != 0:
            return None
        a = self._addr(q)
        return (name, a) if a is not None else None

    def _link_size(self, p: int) -> int:
        b = self.buf
        flags = b[p + 1]
        q = p + 2
        ltype = 0
        if flags & 0x08:
            ltype = b[q]
            q += 1
        if flags & 0x04:
            q += 8
        if flags & 0x10:
            q += 1
        nsz = 1 << (flags & 3)
        nlen = self._u(q, nsz)
        q += nsz + nlen
        if ltype == 0:
            q += self.so
        elif ltype == 1:                          # soft link: length + target path
            q += 2 + self._u(q, 2)
        else:                                     # external / user-defined: length + data
            q += 2 + self._u(q, 2)
        return q - p

    def _symbol_table(self, btree: int, heap: int) -> Iterator[Tuple[str, int]]:
        b = self.buf
        if b[heap:heap + 4] != b"HEAP":
            raise ValueError(f"{self.path}: bad local heap at {heap}")
        data = self._addr(heap + 8 + 2 * self.sl)

        def node(a):
            if b[a:a + 4] == b"SNOD":
                n = self._u(a + 6, 2)
                p = a + 8
                for _ in range(n):
                    off, obj = self._u(p, self.so), self._addr(p + self.so)
                    e = b.find(b"\x00", data + off)
                    yield bytes(b[data + off:e]).decode("utf-8", "replace"), obj
                    p += 2 * self.so + 24
                return
            if b[a:a + 4] != b"TREE":
                raise ValueError(f"{self.path}: bad group B-tree node at {a}")
            used = self._u(a + 6, 2)
            p = a + 8 + 2 * self.so
            for i in range(used):
                p += self.sl                      # key i
                child = self._addr(p)
                p += self.so
                if child is not None:
                    yield from node(child)

        yield from node(btree)

    def _dense_links(self, heap: int) -> Iterator[Tuple[str, int]]:
        b = self.buf
        if b[heap:heap + 4] != b"FRHP":
            raise ValueError(f"{self.path}: bad fractal heap header at {heap}")
        p = heap + 5
        filt_len = self._u(p + 2, 2)
        hflags = b[p + 4]
        p += 5 + 4                                # id length, filter length, flags, max managed size
        p += self.sl + self.so                    # next huge id, huge B-tree address
        p += self.sl + self.so                    # free space, free-space manager address
        p += 4 * self.sl                          # managed space, allocated, iterator offset, #managed
        p += 4 * self.sl                          # huge size/#, tiny size/#
        width = self._u(p, 2)
        start = self._u(p + 2, self.sl)
        max_direct = self._u(p + 2 + self.sl, self.sl)
        max_heap_bits = self._u(p + 2 + 2 * self.sl, 2)
        p += 2 + 2 * self.sl + 2 + 2              # ... starting # rows
        root = self._addr(p)
        cur_rows = self._u(p + self.so, 2)
        if filt_len:
            raise Hdf5Unsupported(f"{self.path}: filtered fractal heap")
        off_bytes = (max_heap_bits + 7) // 8
        has_cksum = bool(hflags & 2)
        max_direct_rows = (max_direct.bit_length() - 1) - (start.bit_length() - 1) + 2

        def row_size(r):
            return start if r < 2 else start << (r - 1)

        def direct(a, size):
            if a is None or b[a:a + 4] != b"FHDB":
                return
            q = a + 5 + self.so + off_bytes + (4 if has_cksum else 0)
            end = a + size
            while q + 4 < end and b[q] == 1:      # link messages back to back; free space is zero-filled
                n = self._link_size(q)
                if q + n > end:
                    break
                link = self._link(q)
                if link:
                    yield link
                q += n

        def indirect(a, rows):
            if a is None or b[a:a + 4] != b"FHIB":
                return
            q = a + 5 + self.so + off_bytes
            drows = min(rows, max_direct_rows)
            for r in range(drows):
                for _ in range(width):
                    yield from direct(self._addr(q), row_size(r))
                    q += self.so
            for r in range(drows, rows):
                child_rows = (row_size(r).bit_length() - 1) - ((start * width).bit_length() - 1) + 1
                for _ in range(width):
                    yield from indirect(self._addr(q), child_rows)
                    q += self.so

        if root is None:
            return
        if cur_rows == 0:
            yield from direct(root, start)
        else:
            yield from indirect(root, cur_rows)


    # ------------------------------------------------------------------ attributes
    def groups(self) -> List[str]:
        return list(self._groups)

    def _parse_dataspace(self, p: int):
        b = self.buf
        ver, rank = b[p], b[p + 1]
        if ver == 2 and b[p + 3] == 2:
            return None                           # null dataspace: no elements
        q = p + (8 if ver == 1 else 4)
        return tuple(self._u(q + i * self.sl, self.sl) for i in range(rank))

    def _global_heap_object(self, addr: int, index: int) -> bytes:
        b = self.buf
        if b[addr:addr + 4] != b"GCOL":
            raise Hdf5Unsupported(f"{self.path}: bad global heap collection at {addr}")
        end = addr + self._u(addr + 8, self.sl)
        q = addr + 8 + self.sl
        while q + 8 + self.sl <= end:
            idx, size = self._u(q, 2), self._u(q + 8, self.sl)
            if idx == 0:
                break
            if idx == index:
                return bytes(b[q + 8 + self.sl:q + 8 + self.sl + size])
            q += 8 + self.sl + ((size + 7) & ~7)
        raise Hdf5Unsupported(f"{self.path}: global heap object {index} not found at {addr}")

    def attrs(self, name: str = "") -> Dict[str, object]:
        """Attributes of the dataset or group `name` ('' = the root group) that live in its object header: numbers and
        fixed-length strings come back as numpy arrays / scalars (bytes for 'S' types, like h5py), variable-length
        strings as str."""
        name = name.strip("/")
        addr = self._objects.get(name, self._groups.get(name))
        if addr is None:
            raise KeyError(name)
        b = self.buf
        out: Dict[str, object] = {}
        for mtype, p, size in self._messages(addr):
            if mtype != 0x0C:
                continue
            ver = b[p]
            nsz, tsz, ssz = self._u(p + 2, 2), self._u(p + 4, 2), self._u(p + 6, 2)
            q = p + 8 + (1 if ver == 3 else 0)
            pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
            aname = bytes(b[q:q + nsz]).split(b"\x00")[0].decode("utf-8", "replace")
            q += pad(nsz)
            tpos = q
            q += pad(tsz)
            shape = self._parse_dataspace(q)
            q += pad(ssz)
            cls, bits0, tsize = b[tpos] & 0x0F, b[tpos + 1], self._u(tpos + 4, 4)
            count = 0 if shape is None else int(np.prod(shape)) if shape else 1
            if cls in (0, 1, 3):
                order = ">" if bits0 & 1 else "<"
                dt = np.dtype(f"S{tsize}") if cls == 3 else np.dtype(f"{order}{'f' if cls == 1 else ('i' if bits0 & 0x08 else 'u')}{tsize}")
                arr = np.frombuffer(bytes(b[q:q + count * tsize]), dtype=dt, count=count)
                out[aname] = arr.reshape(shape) if shape else (arr[0] if count else arr)
            elif cls == 9 and (bits0 & 0x0F) == 1:          # variable-length string(s): {length, global heap address, index}
                vals = []
                for i in range(count):
                    e = q + i * (4 + self.so + 4)
                    gaddr, gidx = self._addr(e + 4), self._u(e + 4 + self.so, 4)
                    vals.append("" if gaddr is None else self._global_heap_object(gaddr, gidx).split(b"\x00")[0].decode("utf-8", "replace"))
                out[aname] = (np.asarray(vals, dtype=object).reshape(shape) if shape else (vals[0] if vals else ""))
            # other classes (compound, references, ...) are skipped
        return out

    # ------------------------------------------------------------------ datasets
    def keys(self) -> List[str]:
        return list(self._objects)

    def __contains__(self, name: str) -> bool:
        return name in self._objects

    def _describe(self, name: str) -> _Dataset:
        b = self.buf
        d = _Dataset()
        for mtype, p, size in self._messages(self._objects[name]):
            if mtype == 0x01:                     # dataspace
                ver, rank, flags = b[p], b[p + 1], b[p + 2]
                q = p + (8 if ver == 1 else 4)
                if ver == 2 and b[p + 3] == 2:    # null dataspace
                    d.shape = (0,)
                else:
                    d.shape = tuple(self._u(q + i * self.sl, self.sl) for i in range(rank))
            elif mtype == 0x03:                   # datatype
                cls, bits0, tsize = b[p] & 0x0F, b[p + 1], self._u(p + 4, 4)
                order = ">" if bits0 & 1 else "<"
                if cls == 0:
                    d.dtype = np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{tsize}")
                elif cls == 1:
                    d.dtype = np.dtype(f"{order}f{tsize}")
                elif cls == 3:
                    d.dtype = np.dtype(f"S{tsize}")
                else:
                    d.dtype = None
            elif mtype == 0x08:                   # data layout
                ver, cls = b[p], b[p + 1]
                if ver == 3 or ver == 4:
                    if cls == 0:
                        n = self._u(p + 2, 2)
                        d.layout = ("compact", p + 4, n)
                    elif cls == 1:
                        d.layout = ("contiguous", self._addr(p + 2), self._u(p + 2 + self.so, self.sl))
                    elif cls == 2 and ver == 3:
                        nd = b[p + 2]
                        bt = self._addr(p + 3)
                        dims = tuple(self._u(p + 3 + self.so + 4 * i, 4) for i in range(nd))
                        d.layout = ("chunked_btree1", bt, dims[:-1])
                    elif cls == 2:
                        cflags, nd, enc = b[p + 2], b[p + 3], b[p + 4]
                        dims = tuple(self._u(p + 5 + enc * i, enc) for i in range(nd))
                        q = p + 5 + enc * nd
                        itype = b[q]
                        q += 1
                        if itype == 1:            # single chunk
                            if cflags & 2:
                                csize, cmask = self._u(q, self.sl), self._u(q + self.sl, 4)
                                q += self.sl + 4
                            else:
                                csize, cmask = None, 0
                            d.layout = ("single_chunk", self._addr(q), dims[:-1], csize, cmask)
                        elif itype == 2:          # implicit: chunks back to back, no filters
                            d.layout = ("implicit", self._addr(q), dims[:-1])
                        elif itype == 3:          # fixed array
                            d.layout = ("fixed_array", self._addr(q + 1), dims[:-1])
                        else:
                            d.layout = ("unsupported", f"chunk index type {itype}")
                    else:
                        d.layout = ("unsupported", f"layout class {cls}")
                else:
                    d.layout = ("unsupported", f"layout message version {ver}")
            elif mtype == 0x0B:                   # filter pipeline
                ver, n = b[p], b[p + 1]
                q = p + (8 if ver == 1 else 2)
                for _ in range(n):
                    fid = self._u(q, 2)
                    q += 2
                    nlen = 0
                    if ver == 1 or fid >= 256:
                        nlen = self._u(q, 2)
                        q += 2
                    q += 2                        # flags
                    ncd = self._u(q, 2)
                    q += 2
                    if ver == 1:
                        nlen = (nlen + 7) & ~7
                    q += nlen
                    cd = tuple(self._u(q + 4 * i, 4) for i in range(ncd))
                    q += 4 * ncd
                    if ver == 1 and ncd & 1:
                        q += 4
                    d.filters.append((fid, cd))
        return d

    def shape(self, name: str) -> Tuple[int, ...]:
        return self._describe(name).shape

    def _unfilter(self, raw: bytes, filters, mask: int, itemsize: int) -> bytes:
        for i in range(len(filters) - 1, -1, -1):  # undo in reverse pipeline order
            if mask & (1 << i):
                continue
            fid, cd = filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cd[0] if cd else itemsize
                n = len(raw) // es
                a = np.frombuffer(raw, dtype=np.uint8, count=n * es).reshape(es, n)
                raw = a.T.tobytes() + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise Hdf5Unsupported(f"{self.path}: filter id {fid}")
        return raw

    def __getitem__(self, name: str) -> np.ndarray:
        if name not in self._objects:
            raise KeyError(name)
        d = self._describe(name)
        if d.dtype is None or d.shape is None or d.layout is None:
            raise Hdf5Unsupported(f"{self.path}:{name}: datatype/dataspace/layout not readable")
        count = int(np.prod(d.shape, dtype=np.int64)) if d.shape else 1
        nbytes = count * d.dtype.itemsize
        kind = d.layout[0]
        b = self.buf
        if kind == "unsupported":
            raise Hdf5Unsupported(f"{self.path}:{name}: {d.layout[1]}")
        if kind == "compact":
            out = np.frombuffer(b[d.layout[1]:d.layout[1] + nbytes], dtype=d.dtype, count=count)
            return out.reshape(d.shape).copy()
        if kind == "contiguous":
            addr = d.layout[1]
            if addr is None or count == 0:        # never written: fill value (zeros)
                return np.zeros(d.shape, dtype=d.dtype)
            return np.frombuffer(b[addr:addr + nbytes], dtype=d.dtype, count=count).reshape(d.shape).copy()
        # ---- chunked
        cdims = d.layout[2]
        if len(cdims) != len(d.shape):
            raise Hdf5Unsupported(f"{self.path}:{name}: chunk rank {len(cdims)} vs dataset rank {len(d.shape)}")
        out = np.zeros(d.shape, dtype=d.dtype)
        csize = int(np.prod(cdims, dtype=np.int64)) * d.dtype.itemsize

        def place(offsets, raw):
            chunk = np.frombuffer(raw, dtype=d.dtype, count=csize // d.dtype.itemsize).reshape(cdims)
            sl_out, sl_in = [], []
            for o, c, n in zip(offsets, cdims, d.shape):
                if o >= n:
                    return
                m = min(c, n - o)
                sl_out.append(slice(o, o + m))
                sl_in.append(slice(0, m))
            out[tuple(sl_out)] = chunk[tuple(sl_in)]

        if kind == "chunked_btree1":
            nd = len(cdims) + 1

            def node(a):
                if a is None:
                    return
                if b[a:a + 4] != b"TREE" or b[a + 4] != 1:
                    raise ValueError(f"{self.path}:{name}: bad chunk B-tree node at {a}")
                level, used = b[a + 5], self._u(a + 6, 2)
                p = a + 8 + 2 * self.so
                ksize = 8 + 8 * nd
                for _ in range(used):
                    size, mask = self._u(p, 4), self._u(p + 4, 4)
                    offs = tuple(self._u(p + 8 + 8 * i, 8) for i in range(nd - 1))
                    child = self._addr(p + ksize)
                    p += ksize + self.so
                    if child is None:
                        continue
                    if level > 0:
                        node(child)
                    else:
                        place(offs, self._unfilter(bytes(b[child:child + size]), d.filters, mask, d.dtype.itemsize))

            node(d.layout[1])
        elif kind == "single_chunk":
            addr, size, mask = d.layout[1], d.layout[3], d.layout[4]
            if addr is not None:
                raw = bytes(b[addr:addr + (size if size is not None else csize)])
                place((0,) * len(cdims), self._unfilter(raw, d.filters if size is not None else [], mask, d.dtype.itemsize))
        elif kind in ("implicit", "fixed_array"):
            grid = [(n + c - 1) // c for n, c in zip(d.shape, cdims)]
            nchunks = int(np.prod(grid, dtype=np.int64))
            entries = []
            if kind == "implicit":
                base = d.layout[1]
                if base is not None:
                    entries = [(base + i * csize, csize, 0) for i in range(nchunks)]
            else:
                entries = self._fixed_array(d.layout[1], nchunks, bool(d.filters), csize)
            for i, ent in enumerate(entries):
                if ent is None or ent[0] is None:
                    continue
                idx, offs = i, []
                for g, c in zip(reversed(grid), reversed(cdims)):
                    offs.append((idx % g) * c)
                    idx //= g
                raw = bytes(b[ent[0]:ent[0] + ent[1]])
                place(tuple(reversed(offs)), self._unfilter(raw, d.filters, ent[2], d.dtype.itemsize))
        return out

    def _fixed_array(self, hdr: Optional[int], nchunks: int, filtered: bool, csize: int):
        b = self.buf
        if hdr is None:
            return []
        if b[hdr:hdr + 4] != b"FAHD":
            raise ValueError(f"{self.path}: bad fixed-array header at {hdr}")
        entry_size, page_bits = b[hdr + 6], b[hdr + 7]
        nent = self._u(hdr + 8, self.sl)
        dblk = self._addr(hdr + 8 + self.sl)
        if dblk is None:
            return []
        if b[dblk:dblk + 4] != b"FADB":
            raise ValueError(f"{self.path}: bad fixed-array data block at {dblk}")
        p = dblk + 6 + self.so
        if nent > (1 << page_bits):
            raise Hdf5Unsupported(f"{self.path}: paged fixed-array chunk index")
        out = []
        for _ in range(min(nent, nchunks)):
            a = self._addr(p)
            if filtered:
                sz_bytes = entry_size - self.so - 4
                out.append((a, self._u(p + self.so, sz_bytes), self._u(p + self.so + sz_bytes, 4)))
            else:
                out.append((a, csize, 0))
            p += entry_size
        return out


def read_hdf5(path: str, skip_unsupported: bool = True) -> Dict[str, np.ndarray]:
    """Every readable dataset of the file as an ndarray (native byte order)."""
    out: Dict[str, np.ndarray] = {}
    with Hdf5File(path) as f:
        for k in f.keys():
            try:
                a = f[k]
            except Hdf5Unsupported:
                if skip_unsupported:
                    continue
                raise
            if a.dtype.kind in "iuf" and not a.dtype.isnative:
                a = a.astype(a.dtype.newbyteorder("="))
            out[k] = a
    return out


# ---------------------------------------------------------------------------------------------------
# Writer: one contiguous float32/float64 dataset in the root group (what the reference's `save_h5`
# branches produce with h5py: dataset 'data' for the splits, 'pred' for predictions,
# climsim_utils/data_utils.py:906-925).  Oldest format variants, which every HDF5 library reads:
# superblock 0, symbol-table root group, object headers version 1.
_UNDEF = (1 << 64) - 1


def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * (-len(b) % 8)


def _ohdr_v1(messages) -> bytes:
    body = b""
    for mtype, data in messages:
        data = _pad8(data)
        body += struct.pack("<HHB3x", mtype, len(data), 0) + data
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


def write_hdf5_dataset(path: str, name: str, array) -> None:
    a = np.ascontiguousarray(array)
    if a.dtype not in (np.float32, np.float64):
        raise Hdf5Unsupported(f"write_hdf5_dataset: dtype {a.dtype} (float32 / float64 only)")
    a = a.astype(a.dtype.newbyteorder("<"), copy=False)
    if a.dtype.itemsize == 8:
        ftype = struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    else:
        ftype = struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    nm = name.encode("utf-8") + b"\x00"
    heap_data = _pad8(b"\x00") + _pad8(nm)
    heap_data += b"\x00" * (-len(heap_data) % 16)
    GROUP_LEAF_K, GROUP_INT_K = 4, 16
    # ---- file map
    a_root = 96
    root_ohdr_len = 16 + 8 + 16
    a_btree = a_root + root_ohdr_len
    btree_len = 24 + (2 * GROUP_INT_K + 1) * 8 + 2 * GROUP_INT_K * 8
    a_heap = a_btree + btree_len
    a_heapdata = a_heap + 32
    a_snod = a_heapdata + len(heap_data)
    snod_len = 8 + 2 * GROUP_LEAF_K * 40
    a_dset = a_snod + snod_len
    dataspace = struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", n) for n in a.shape)
    fill = struct.pack("<BBBB", 2, 2, 2, 0)              # version 2, late allocation, fill written if set, undefined
    a_data_placeholder = 0
    layout = struct.pack("<BBQQ", 3, 1, a_data_placeholder, a.nbytes)
    dset = _ohdr_v1([(0x01, dataspace), (0x03, ftype), (0x05, fill), (0x08, layout)])
    a_data = (a_dset + len(dset) + 7) & ~7
    layout = struct.pack("<BBQQ", 3, 1, a_data, a.nbytes)
    dset = _ohdr_v1([(0x01, dataspace), (0x03, ftype), (0x05, fill), (0x08, layout)])
    eof = a_data + a.nbytes
    # ---- pieces
    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, GROUP_LEAF_K, GROUP_INT_K, 0)
    sb += struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
    sb += struct.pack("<QQII", 0, a_root, 1, 0) + struct.pack("<QQ", a_btree, a_heap)
    assert len(sb) == 96
    root = _ohdr_v1([(0x11, struct.pack("<QQ", a_btree, a_heap))])
    assert len(root) == root_ohdr_len
    bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, _UNDEF, _UNDEF) + struct.pack("<QQQ", 0, a_snod, 8)
    bt += b"\x00" * (btree_len - len(bt))
    heap = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, a_heapdata)     # free-list head 1 = none
    snod = b"SNOD" + struct.pack("<BBH", 1, 0, 1) + struct.pack("<QQII16x", 8, a_dset, 0, 0)
    snod += b"\x00" * (snod_len - len(snod))
    with open(path, "wb") as f:
        f.write(sb + root + bt + heap + heap_data + snod + dset)
        f.write(b"\x00" * (a_data - (a_dset + len(dset))))
        f.write(a.tobytes())


# ---------------------------------------------------------------------------------------------- tree writer
# Groups, datasets and attributes - what a Keras `.h5` checkpoint consists of (model.save / save_weights through h5py:
# step2_retrain.py:253-261).  Same conservative choices as `write_hdf5_dataset`: superblock 0, version-1 object headers,
# symbol-table groups (one v1 B-tree leaf + one symbol node per group; the file-wide "group leaf node K" is sized for the
# largest group), contiguous little-endian datasets, version-1 attribute messages with fixed-length strings - the forms
# every HDF5 library version reads.
ATTRS = "@attrs"


def _dtype_message(a: np.ndarray) -> bytes:
    if a.dtype.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, max(a.dtype.itemsize, 1))          # string, null-padded, ASCII
    if a.dtype == np.float32:
        return struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    if a.dtype == np.float64:
        return struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    if a.dtype.kind in "iu" and a.dtype.itemsize in (1, 2, 4, 8):
        return struct.pack("<BBBBI", 0x10, 0x08 if a.dtype.kind == "i" else 0x00, 0, 0, a.dtype.itemsize) + struct.pack(
            "<HH", 0, 8 * a.dtype.itemsize)
    raise Hdf5Unsupported(f"write_hdf5_tree: dtype {a.dtype}")


def _as_le_array(v) -> np.ndarray:
    if isinstance(v, str):
        v = v.encode("utf-8")
    if isinstance(v, bytes):
        return np.asarray(v, dtype=f"S{max(len(v), 1)}")
    a = np.asarray(v)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf-8")
    if a.dtype.kind == "O":
        raise Hdf5Unsupported("write_hdf5_tree: object arrays")
    if a.dtype.kind == "b":
        a = a.astype(np.uint8)
    if a.dtype.kind in "fiu":
        a = a.astype(a.dtype.newbyteorder("<"), copy=False)
    return a if a.flags.c_contiguous else np.ascontiguousarray(a)      # (ascontiguousarray would turn a scalar into shape (1,))


def _dataspace_message(shape) -> bytes:
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(n)) for n in shape)


def _attribute_message(name: str, value) -> bytes:
    a = _as_le_array(value)
    nm = name.encode("utf-8") + b"\x00"
    dt, sp = _dtype_message(a), _dataspace_message(a.shape)
    return (struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + a.tobytes())


class _TreeWriter:
    INT_K = 16

    def __init__(self, tree: dict):
        self.tree = tree
        self.buf = bytearray(96)                  # superblock goes here at the end
        self.leaf_k = max(4, (self._max_children(tree) + 1) // 2)

    def _max_children(self, node) -> int:
        kids = [k for k in node if k != ATTRS]
        return max([len(kids)] + [self._max_children(node[k]) for k in kids if isinstance(node[k], dict)])

    def _alloc(self, data: bytes) -> int:
        self.buf += b"\x00" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def _dataset(self, value) -> int:
        a = _as_le_array(value)
        data_addr = self._alloc(a.tobytes() if a.size else b"")
        fill = struct.pack("<BBBB", 2, 2, 2, 0)
        layout = struct.pack("<BBQQ", 3, 1, data_addr if a.size else _UNDEF, a.nbytes)
        return self._alloc(_ohdr_v1([(0x01, _dataspace_message(a.shape)), (0x03, _dtype_message(a)), (0x05, fill), (0x08, layout)]))

    def _group(self, node: dict):
        """-> (object header address, b-tree address, local heap address)"""
        names = sorted((k for k in node if k != ATTRS), key=lambda s: s.encode("utf-8"))
        child = {k: (self._group(node[k])[0] if isinstance(node[k], dict) else self._dataset(node[k])) for k in names}
        heap_data = bytearray(_pad8(b"\x00"))
        offs = {}
        for k in names:
            offs[k] = len(heap_data)
            heap_data += _pad8(k.encode("utf-8") + b"\x00")
        heap_data += b"\x00" * (-len(heap_data) % 16)
        heap_data_addr = self._alloc(bytes(heap_data))
        heap_addr = self._alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, heap_data_addr))      # free-list head 1 = none
        snod = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(names)))
        for k in names:
            snod += struct.pack("<QQII16x", offs[k], child[k], 0, 0)
        snod += b"\x00" * (8 + 2 * self.leaf_k * 40 - len(snod))
        snod_addr = self._alloc(bytes(snod))
        bt_len = 24 + (2 * self.INT_K + 1) * 8 + 2 * self.INT_K * 8
        if names:
            bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, _UNDEF, _UNDEF) + struct.pack("<QQQ", 0, snod_addr, offs[names[-1]])
        else:
            bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 0, _UNDEF, _UNDEF)
        bt_addr = self._alloc(bt + b"\x00" * (bt_len - len(bt)))
        msgs = [(0x11, struct.pack("<QQ", bt_addr, heap_addr))]
        msgs += [(0x0C, _attribute_message(k, v)) for k, v in node.get(ATTRS, {}).items()]
        return self._alloc(_ohdr_v1(msgs)), bt_addr, heap_addr

    def finish(self) -> bytes:
        root, bt, heap = self._group(self.tree)
        eof = len(self.buf)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.leaf_k, self.INT_K, 0)
        sb += struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
        sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", bt, heap)
        assert len(sb) == 96
        self.buf[0:96] = sb
        return bytes(self.buf)


def write_hdf5_tree(path: str, tree: dict) -> None:
    """Write nested groups: `tree` maps names to arrays (datasets) or dicts (groups); the key '@attrs' of a dict holds that
    group's attributes (numbers, numeric arrays, str / bytes, arrays of bytes).  Written atomically (tmp + replace)."""
    import os
    data = _TreeWriter(tree).finish()
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)
