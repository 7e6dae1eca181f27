"""HDF5 with nothing but NumPy: the subset of the format that 10x Genomics
count matrices (CellRanger ``*.h5``: a group per genome holding ``data``,
``indices``, ``indptr``, ``shape``, ``barcodes``, ``gene_names``), generic
sparse ``.h5`` files and the reference's own ``.sparse.h5`` cache use.

The reference reads and writes these files through PyTables
(``scvae/data/loaders.py:651-676, 725-745``; ``scvae/data/internal_io.py``);
neither PyTables nor h5py exists in this image, and the files are simple enough
to read from their published layout (HDF5 File Format Specification 1.x/2.0/3.0):

reader -- superblock versions 0-3; object headers versions 1 and 2 (with
    continuation blocks); old-style groups (symbol table: B-tree v1 + local
    heap) and new-style groups with compact link messages; data layouts
    compact / contiguous / chunked (B-tree v1 index; single-chunk and implicit
    index of layout version 4); filters deflate, shuffle and fletcher32;
    integer, IEEE float, fixed-length string, enum-over-integer (h5py's bool)
    and variable-length string (global heap) types; attribute messages
    versions 1-3.  Anything else raises ``Hdf5Error`` naming the feature.
writer -- superblock 0, version-1 object headers, symbol-table groups,
    contiguous datasets, scalar / 1-D attributes: what ``.sparse.h5`` needs,
    readable by libhdf5 (checked against h5py when the fixtures were made,
    ``tests/golden/make_hdf5_fixtures.py``).
"""

import mmap
import struct
import zlib

import numpy

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEFINED = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


# --------------------------------------------------------------------------
# reading
# --------------------------------------------------------------------------

class File:
    """``File(path)`` -- use as a context manager; ``file.root`` is a Group."""

    def __init__(self, path):
        self.path = path
        self._handle = open(path, "rb")
        try:
            self.buf = mmap.mmap(self._handle.fileno(), 0,
                                 access=mmap.ACCESS_READ)
        except ValueError:
            self._handle.close()
            raise Hdf5Error("`{}` is empty.".format(path))
        self._global_heaps = {}
        self._read_superblock()

    def close(self):
        self.buf.close()
        self._handle.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- primitives -----------------------------------------------------------
    def u(self, offset, size):
        return int.from_bytes(self.buf[offset:offset + size], "little")

    def address(self, offset):
        value = self.u(offset, self.O)
        if value == (1 << (8 * self.O)) - 1:
            return UNDEFINED
        return value + self.base

    def _read_superblock(self):
        offset, size = 0, len(self.buf)
        while offset + 8 <= size and self.buf[offset:offset + 8] != SIGNATURE:
            offset = 512 if offset == 0 else offset * 2
        if offset + 8 > size:
            raise Hdf5Error("`{}` is not an HDF5 file.".format(self.path))
        version = self.buf[offset + 8]
        self.base = 0
        if version in (0, 1):
            self.O, self.L = self.buf[offset + 13], self.buf[offset + 14]
            self.group_leaf_k = self.u(offset + 16, 2)
            self.group_internal_k = self.u(offset + 18, 2)
            p = offset + 24 + (4 if version == 1 else 0)
            self.base = self.u(p, self.O)
            p += 4 * self.O
            # root group symbol table entry: name offset, object header address
            root = self.address(p + self.O)
        elif version in (2, 3):
            self.O, self.L = self.buf[offset + 9], self.buf[offset + 10]
            self.group_leaf_k, self.group_internal_k = 4, 16
            p = offset + 12
            self.base = self.u(p, self.O)
            root = self.address(p + 3 * self.O)
        else:
            raise Hdf5Error("superblock version {}".format(version))
        self.root = Group(self, root, "/")

    def global_heap_object(self, collection, index):
        if collection not in self._global_heaps:
            b = self.buf
            if b[collection:collection + 4] != b"GCOL":
                raise Hdf5Error("global heap signature")
            size = self.u(collection + 8, self.L)
            objects, p = {}, collection + 8 + self.L
            end = collection + size
            while p + 8 + self.L <= end:
                i = self.u(p, 2)
                n = self.u(p + 8, self.L)
                if i == 0:
                    break
                start = p + 8 + self.L
                objects[i] = bytes(b[start:start + n])
                p = start + (n + 7) // 8 * 8
            self._global_heaps[collection] = objects
        return self._global_heaps[collection][index]


class Node:
    """An object header: its messages as (type, bytes) pairs."""

    def __init__(self, file, address, name):
        self.file, self.address, self.name = file, address, name
        self.messages = []
        self._attributes = None
        b = file.buf
        if b[address:address + 4] == b"OHDR":
            self._read_v2(address)
        elif b[address] == 1:
            self._read_v1(address)
        else:
            raise Hdf5Error("object header version {}".format(b[address]))

    def _read_v1(self, address):
        f = self.file
        count = f.u(address + 2, 2)
        size = f.u(address + 8, 4)
        blocks = [(address + 16, size)]
        while blocks and len(self.messages) < count:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(self.messages) < count:
                kind, length = f.u(p, 2), f.u(p + 2, 2)
                data = bytes(f.buf[p + 8:p + 8 + length])
                p += 8 + length
                if kind == 0x10:     # continuation
                    blocks.append((f.address_of(data, 0),
                                   int.from_bytes(data[f.O:f.O + f.L], "little")))
                self.messages.append((kind, data))

    def _read_v2(self, address):
        f = self.file
        flags = f.buf[address + 5]
        p = address + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        width = 1 << (flags & 3)
        size = f.u(p, width)
        p += width
        blocks = [(p, size)]
        tracked = 2 if flags & 4 else 0
        while blocks:
            p, n = blocks.pop(0)
            end = p + n
            while p + 4 + tracked <= end:
                kind, length = f.buf[p], f.u(p + 1, 2)
                start = p + 4 + tracked
                data = bytes(f.buf[start:start + length])
                p = start + length
                if kind == 0x10:
                    where = f.address_of(data, 0)
                    n2 = int.from_bytes(data[f.O:f.O + f.L], "little")
                    if f.buf[where:where + 4] != b"OCHK":
                        raise Hdf5Error("object header continuation")
                    blocks.append((where + 4, n2 - 8))   # minus signature, checksum
                elif kind != 0:
                    self.messages.append((kind, data))

    def message(self, kind):
        for k, data in self.messages:
            if k == kind:
                return data
        return None

    @property
    def attributes(self):
        if self._attributes is None:
            self._attributes = {}
            for kind, data in self.messages:
                if kind == 0x0C:
                    name, value = _attribute(self.file, data)
                    self._attributes[name] = value
        return self._attributes


def _address_of(self, data, offset):
    value = int.from_bytes(data[offset:offset + self.O], "little")
    if value == (1 << (8 * self.O)) - 1:
        return UNDEFINED
    return value + self.base


File.address_of = _address_of


class Group(Node):
    def links(self):
        """{name: object header address} of the group's members."""
        f = self.file
        found = {}
        table = self.message(0x11)
        if table is not None:
            tree = f.address_of(table, 0)
            heap = f.address_of(table, f.O)
            if f.buf[heap:heap + 4] != b"HEAP":
                raise Hdf5Error("local heap signature")
            segment = f.address(heap + 8 + 2 * f.L)
            self._walk_group_tree(tree, segment, found)
        for kind, data in self.messages:
            if kind == 0x06:
                name, where = _link(f, data)
                if where is not None:
                    found[name] = where
            elif kind == 0x02:
                flags = data[1]
                p = 2 + (8 if flags & 1 else 0)
                if f.address_of(data, p) != UNDEFINED:
                    raise Hdf5Error(
                        "group `{}` stores its links in a fractal heap (more "
                        "than eight members written with libver='latest'): "
                        "not supported".format(self.name))
        return found

    def _walk_group_tree(self, node, segment, found):
        f, b = self.file, self.file.buf
        if node == UNDEFINED:
            return
        if b[node:node + 4] != b"TREE" or b[node + 4] != 0:
            raise Hdf5Error("group B-tree node")
        level, used = b[node + 5], f.u(node + 6, 2)
        p = node + 8 + 2 * f.O + f.L          # past the siblings and key 0
        for _ in range(used):
            child = f.address(p)
            p += f.O + f.L
            if level > 0:
                self._walk_group_tree(child, segment, found)
                continue
            if b[child:child + 4] != b"SNOD":
                raise Hdf5Error("symbol table node")
            entry = child + 8
            for _ in range(f.u(child + 6, 2)):
                name_at = segment + f.u(entry, f.O)
                end = b.find(b"\x00", name_at)
                found[bytes(b[name_at:end]).decode("utf-8")] = f.address(
                    entry + f.O)
                entry += 2 * f.O + 24

    def __contains__(self, name):
        return name in self.links()

    def keys(self):
        return sorted(self.links())

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            links = node.links()
            if part not in links:
                raise KeyError(path)
            prefix = "" if node.name == "/" else node.name
            node = _open(self.file, links[part], prefix + "/" + part)
        return node

    def walk(self):
        """Every node below this group, depth first, groups before their
        members (PyTables' ``walk_nodes``)."""
        for name in self.keys():
            child = self[name]
            yield child
            if isinstance(child, Group):
                for below in child.walk():
                    yield below


def _open(file, address, name):
    node = Node(file, address, name)
    kinds = {k for k, _ in node.messages}
    cls = Dataset if (0x08 in kinds and 0x03 in kinds) else Group
    node.__class__ = cls
    return node


def _link(f, data):
    flags = data[1]
    p = 2
    kind = 0
    if flags & 8:
        kind = data[p]
        p += 1
    if flags & 4:
        p += 8
    if flags & 16:
        p += 1
    width = 1 << (flags & 3)
    n = int.from_bytes(data[p:p + width], "little")
    p += width
    name = data[p:p + n].decode("utf-8")
    p += n
    if kind != 0:       # soft / external links: not followed
        return name, None
    return name, f.address_of(data, p)


# -- datatypes ---------------------------------------------------------------

class _Type:
    def __init__(self, dtype, size, vlen_string=False, consumed=0):
        self.dtype, self.size, self.vlen_string = dtype, size, vlen_string
        self.consumed = consumed


def _datatype(data, offset=0):
    cls, version = data[offset] & 15, data[offset] >> 4
    bits = data[offset + 1:offset + 4]
    size = int.from_bytes(data[offset + 4:offset + 8], "little")
    order = ">" if bits[0] & 1 else "<"
    if cls == 0:
        dtype = numpy.dtype("{}{}{}".format(order, "i" if bits[0] & 8 else "u",
                                            size))
        return _Type(dtype, size, consumed=12)
    if cls == 1:
        return _Type(numpy.dtype("{}f{}".format(order, size)), size,
                     consumed=20)
    if cls == 3:
        return _Type(numpy.dtype("S{}".format(size)), size, consumed=8)
    if cls == 8:          # enumeration over an integer (h5py's bool)
        base = _datatype(data, offset + 8)
        return _Type(base.dtype, size, consumed=len(data) - offset)
    if cls == 9:
        if (bits[0] & 15) == 1:
            return _Type(numpy.dtype(object), size, vlen_string=True,
                         consumed=len(data) - offset)
        raise Hdf5Error("variable-length sequences are not supported")
    names = {2: "time", 4: "bit field", 5: "opaque", 6: "compound",
             7: "reference", 10: "array"}
    raise Hdf5Error("datatype class `{}` is not supported".format(
        names.get(cls, cls)))


def _dataspace(f, data, offset=0):
    version, rank, flags = data[offset], data[offset + 1], data[offset + 2]
    p = offset + (8 if version == 1 else 4)
    if version == 2 and data[offset + 3] == 2:
        return None       # null dataspace
    return tuple(int.from_bytes(data[p + i * f.L:p + (i + 1) * f.L], "little")
                 for i in range(rank))


def _decode(f, t, raw, shape):
    count = int(numpy.prod(shape)) if shape else 1
    if t.vlen_string:
        out = numpy.empty(count, dtype=object)
        step = 4 + f.O + 4
        for i in range(count):
            p = i * step
            n = int.from_bytes(raw[p:p + 4], "little")
            where = int.from_bytes(raw[p + 4:p + 4 + f.O], "little") + f.base
            index = int.from_bytes(raw[p + 4 + f.O:p + step], "little")
            out[i] = (f.global_heap_object(where, index)[:n].decode("utf-8")
                      if n else "")
        return out.reshape(shape)
    array = numpy.frombuffer(raw, dtype=t.dtype, count=count)
    if t.dtype.byteorder == ">":
        array = array.astype(t.dtype.newbyteorder("<"))
    return array.reshape(shape).copy()


def _attribute(f, data):
    version = data[0]
    n_name = int.from_bytes(data[2:4], "little")
    n_type = int.from_bytes(data[4:6], "little")
    n_space = int.from_bytes(data[6:8], "little")
    p = 8 + (1 if version == 3 else 0)
    pad = (lambda n: (n + 7) // 8 * 8) if version == 1 else (lambda n: n)
    name = data[p:p + n_name].split(b"\x00")[0].decode("utf-8")
    p += pad(n_name)
    try:
        t = _datatype(data, p)
    except Hdf5Error:
        return name, None
    shape = _dataspace(f, data, p + pad(n_type))
    p += pad(n_type) + pad(n_space)
    if shape is None:
        return name, None
    count = int(numpy.prod(shape)) if shape else 1
    step = (4 + f.O + 4) if t.vlen_string else t.size
    value = _decode(f, t, data[p:p + count * step], shape)
    if value.dtype.kind == "S":
        text = numpy.char.decode(value, "utf-8")
        value = text
    if shape == ():
        value = value.reshape(()).item()
    return name, value


class Dataset(Node):
    @property
    def type(self):
        return _datatype(self.message(0x03))

    @property
    def shape(self):
        return _dataspace(self.file, self.message(0x01))

    @property
    def dtype(self):
        return self.type.dtype

    def _filters(self):
        data = self.message(0x0B)
        if data is None:
            return []
        version, count = data[0], data[1]
        p = 8 if version == 1 else 2
        filters = []
        for _ in range(count):
            ident = int.from_bytes(data[p:p + 2], "little")
            p += 2
            n_name = 0
            if version == 1 or ident >= 256:
                n_name = int.from_bytes(data[p:p + 2], "little")
                p += 2
            p += 2   # flags
            n_values = int.from_bytes(data[p:p + 2], "little")
            p += 2
            if version == 1:
                n_name = (n_name + 7) // 8 * 8
            p += n_name
            values = [int.from_bytes(data[p + 4 * i:p + 4 * i + 4], "little")
                      for i in range(n_values)]
            p += 4 * n_values
            if version == 1 and n_values % 2:
                p += 4
            filters.append((ident, values))
        return filters

    def _unfilter(self, raw, filters, mask, element_size):
        for index in range(len(filters) - 1, -1, -1):
            if mask & (1 << index):
                continue
            ident, values = filters[index]
            if ident == 1:
                raw = zlib.decompress(raw)
            elif ident == 2:
                size = values[0] if values else element_size
                n = len(raw) // size
                raw = (numpy.frombuffer(raw[:n * size], dtype=numpy.uint8)
                       .reshape(size, n).T.tobytes() + raw[n * size:])
            elif ident == 3:
                raw = raw[:-4]
            else:
                raise Hdf5Error(
                    "filter {} (only deflate, shuffle and fletcher32 are "
                    "supported)".format(ident))
        return raw

    def read(self):
        """The whole dataset as a NumPy array (strings: fixed ``S``, or objects
        for variable-length strings)."""
        f = self.file
        t = self.type
        shape = self.shape
        if shape is None:
            return None
        layout = self.message(0x08)
        version = layout[0]
        step = (4 + f.O + 4) if t.vlen_string else t.size
        count = int(numpy.prod(shape)) if shape else 1
        if version in (1, 2):
            rank, cls = layout[1], layout[2]
            p = 8
            where = None
            if cls != 0:
                where = f.address_of(layout, p)
                p += f.O
            dims = [int.from_bytes(layout[p + 4 * i:p + 4 * i + 4], "little")
                    for i in range(rank)]
            p += 4 * rank
            if cls == 0:
                n = int.from_bytes(layout[p:p + 4], "little")
                return _decode(f, t, layout[p + 4:p + 4 + n], shape)
            if cls == 1:
                return self._contiguous(where, count * step, t, shape)
            return self._chunked(where, dims[:-1], t, shape, step)
        if version not in (3, 4):
            raise Hdf5Error("data layout version {}".format(version))
        cls = layout[1]
        if cls == 0:
            n = int.from_bytes(layout[2:4], "little")
            return _decode(f, t, layout[4:4 + n], shape)
        if cls == 1:
            return self._contiguous(f.address_of(layout, 2), count * step, t,
                                    shape)
        if cls != 2:
            raise Hdf5Error("data layout class {}".format(cls))
        if version == 3:
            rank = layout[2]
            where = f.address_of(layout, 3)
            p = 3 + f.O
            dims = [int.from_bytes(layout[p + 4 * i:p + 4 * i + 4], "little")
                    for i in range(rank)]
            return self._chunked(where, dims[:-1], t, shape, step)
        # version 4: flags, rank, dimension size width, dims, index type
        flags, rank, width = layout[2], layout[3], layout[4]
        p = 5
        dims = [int.from_bytes(layout[p + width * i:p + width * (i + 1)],
                               "little") for i in range(rank)]
        p += width * rank
        index = layout[p]
        p += 1
        chunk = dims[:-1]
        if index == 1:      # single chunk
            size, mask = count * step, 0
            if flags & 2:
                size = int.from_bytes(layout[p:p + f.L], "little")
                mask = int.from_bytes(layout[p + f.L:p + f.L + 4], "little")
                p += f.L + 4
            where = f.address_of(layout, p)
            out = numpy.zeros(shape, dtype=t.dtype)
            self._place(out, (0,) * len(shape), chunk, bytes(
                f.buf[where:where + size]), mask, t, step)
            return out
        if index == 2:      # implicit: chunks one after the other, unfiltered
            where = f.address_of(layout, p)
            out = numpy.zeros(shape, dtype=t.dtype)
            chunk_bytes = int(numpy.prod(chunk)) * step
            grid = [-(-s // c) for s, c in zip(shape, chunk)]
            for i, position in enumerate(numpy.ndindex(*grid)):
                offset = tuple(a * c for a, c in zip(position, chunk))
                start = where + i * chunk_bytes
                self._place(out, offset, chunk, bytes(
                    f.buf[start:start + chunk_bytes]), ~0, t, step)
            return out
        raise Hdf5Error(
            "chunk index type {} (fixed / extensible array, B-tree v2: files "
            "written with libver='latest') is not supported".format(index))

    def _contiguous(self, where, size, t, shape):
        if where == UNDEFINED:      # never written: the fill value (zero)
            return numpy.zeros(shape, dtype=t.dtype)
        return _decode(self.file, t, bytes(self.file.buf[where:where + size]),
                       shape)

    def _place(self, out, offset, chunk, raw, mask, t, step):
        filters = self._filters()
        raw = self._unfilter(raw, filters, mask, step)
        if t.vlen_string:
            block = _decode(self.file, t, raw, tuple(chunk))
        else:
            block = numpy.frombuffer(raw, dtype=t.dtype,
                                     count=int(numpy.prod(chunk))).reshape(chunk)
        index, part = [], []
        for o, c, s in zip(offset, chunk, out.shape):
            n = min(c, s - o)
            if n <= 0:
                return
            index.append(slice(o, o + n))
            part.append(slice(0, n))
        out[tuple(index)] = block[tuple(part)]

    def _chunked(self, tree, chunk, t, shape, step):
        out = (numpy.empty(shape, dtype=object) if t.vlen_string
               else numpy.zeros(shape, dtype=t.dtype.newbyteorder("=")
                                if t.dtype.byteorder == ">" else t.dtype))
        if t.vlen_string:
            out[...] = ""
        if tree != UNDEFINED:
            self._walk_chunks(tree, len(shape), chunk, out, t, step)
        return out

    def _walk_chunks(self, node, rank, chunk, out, t, step):
        f, b = self.file, self.file.buf
        if b[node:node + 4] != b"TREE" or b[node + 4] != 1:
            raise Hdf5Error("chunk B-tree node")
        level, used = b[node + 5], f.u(node + 6, 2)
        key = 8 + 8 * (rank + 1)
        p = node + 8 + 2 * f.O
        for _ in range(used):
            size, mask = f.u(p, 4), f.u(p + 4, 4)
            offset = tuple(f.u(p + 8 + 8 * i, 8) for i in range(rank))
            child = f.address(p + key)
            p += key + f.O
            if level > 0:
                self._walk_chunks(child, rank, chunk, out, t, step)
            else:
                self._place(out, offset, chunk, bytes(b[child:child + size]),
                            mask, t, step)


# --------------------------------------------------------------------------
# writing
# --------------------------------------------------------------------------

def _pad8(data):
    return data + b"\x00" * (-len(data) % 8)


def _type_message(dtype):
    dtype = numpy.dtype(dtype)
    size = dtype.itemsize
    if dtype.kind in "iu":
        bits = 8 if dtype.kind == "i" else 0
        return (bytes([0x10, bits, 0, 0]) + struct.pack("<I", size)
                + struct.pack("<HH", 0, 8 * size))
    if dtype.kind == "f" and size in (4, 8):
        if size == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        else:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        return (bytes([0x11, 0x20, 8 * size - 1, 0]) + struct.pack("<I", size)
                + props)
    if dtype.kind == "S":
        # null-padded ASCII (what h5py writes for numpy ``S`` arrays)
        return bytes([0x13, 0x01, 0, 0]) + struct.pack("<I", max(size, 1))
    raise Hdf5Error("cannot write dtype {}".format(dtype))


def _space_message(shape):
    if shape == ():
        return bytes([1, 0, 0, 0, 0, 0, 0, 0])
    data = bytes([1, len(shape), 1, 0, 0, 0, 0, 0])
    dims = b"".join(struct.pack("<Q", n) for n in shape)
    return data + dims + dims      # current and maximum dimensions


def _as_stored(value):
    if isinstance(value, str):
        value = numpy.array(value.encode("utf-8") or b"\x00", dtype="S")
    value = numpy.asarray(value)
    if value.dtype.kind == "U":
        value = numpy.char.encode(value, "utf-8")
    if value.dtype.kind == "b":
        value = value.astype(numpy.uint8)
    if value.dtype.kind == "S" and value.dtype.itemsize == 0:
        value = value.astype("S1")
    if value.dtype.byteorder == ">":
        value = value.astype(value.dtype.newbyteorder("<"))
    return numpy.asarray(value, order="C")      # (ascontiguousarray would make a scalar 1-D)


def _message(kind, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHBBBB", kind, len(data), flags, 0, 0, 0) + data


def _attribute_message(name, value):
    value = _as_stored(value)
    encoded = name.encode("utf-8") + b"\x00"
    t, s = _type_message(value.dtype), _space_message(value.shape)
    head = struct.pack("<BBHHH", 1, 0, len(encoded), len(t), len(s))
    return _message(0x0C, head + _pad8(encoded) + _pad8(t) + _pad8(s)
                    + value.tobytes())


class WriterGroup:
    def __init__(self, attrs=None):
        self.members = {}
        self.attrs = dict(attrs or {})

    def create_group(self, name, attrs=None):
        group = WriterGroup(attrs)
        self.members[name] = group
        return group

    def create_dataset(self, name, array, attrs=None):
        self.members[name] = (_as_stored(array), dict(attrs or {}))


class Writer:
    """``w = Writer(); w.root.create_group(...).create_dataset(...); w.save(path)``"""

    LEAF_K, INTERNAL_K = 4, 16

    def __init__(self):
        self.root = WriterGroup()

    def save(self, path):
        self.out = bytearray(96)          # the superblock comes last
        tree, heap, header = self._group(self.root)
        end = len(self.out)
        sb = bytearray()
        sb += SIGNATURE
        sb += bytes([0, 0, 0, 0, 0, 8, 8, 0])
        sb += struct.pack("<HHI", self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEFINED, end, UNDEFINED)
        sb += struct.pack("<QQII", 0, header, 1, 0)
        sb += struct.pack("<QQ", tree, heap)
        self.out[:96] = sb
        with open(path, "wb") as handle:
            handle.write(bytes(self.out))

    def _allocate(self, data):
        self.out += b"\x00" * (-len(self.out) % 8)
        address = len(self.out)
        self.out += data
        return address

    def _header(self, messages):
        body = b"".join(messages)
        head = struct.pack("<BBHII", 1, 0, len(messages), 1, len(body))
        return self._allocate(head + b"\x00" * 4 + body)

    def _dataset(self, array, attrs):
        where = self._allocate(array.tobytes()) if array.size else UNDEFINED
        layout = bytes([3, 1]) + struct.pack("<QQ", where, array.nbytes)
        messages = [_message(0x01, _space_message(array.shape)),
                    _message(0x03, _type_message(array.dtype), flags=1),
                    # fill value: allocate early, never write, undefined
                    _message(0x05, bytes([2, 1, 2, 0])),
                    _message(0x08, layout)]
        messages += [_attribute_message(k, v) for k, v in attrs.items()]
        return self._header(messages)

    def _group(self, group):
        names = sorted(group.members, key=lambda n: n.encode("utf-8"))
        if len(names) > 2 * self.LEAF_K * 2 * self.INTERNAL_K:
            raise Hdf5Error("too many members in one group")
        # local heap data: "" at offset 0, then the names
        heap_data = bytearray(8)
        offsets = {}
        for name in names:
            offsets[name] = len(heap_data)
            heap_data += _pad8(name.encode("utf-8") + b"\x00")
        free_at = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)      # one free block, the last
        addresses = {}
        for name in names:
            member = group.members[name]
            if isinstance(member, WriterGroup):
                addresses[name] = self._group(member)
            else:
                addresses[name] = (None, None, self._dataset(*member))
        segment = self._allocate(bytes(heap_data))
        heap = self._allocate(b"HEAP" + bytes([0, 0, 0, 0]) + struct.pack(
            "<QQQ", len(heap_data), free_at, segment))
        # symbol table nodes of up to 2 K entries, one B-tree node over them
        per = 2 * self.LEAF_K
        nodes, keys = [], [0]
        for start in range(0, len(names), per):
            part = names[start:start + per]
            body = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(part)))
            for name in part:
                tree, sub_heap, header = addresses[name]
                if tree is None:
                    body += struct.pack("<QQII", offsets[name], header, 0, 0)
                    body += b"\x00" * 16
                else:
                    body += struct.pack("<QQII", offsets[name], header, 1, 0)
                    body += struct.pack("<QQ", tree, sub_heap)
            body += b"\x00" * (8 + per * 40 - len(body))
            nodes.append(self._allocate(bytes(body)))
            keys.append(offsets[part[-1]] if part else 0)
        body = bytearray(b"TREE" + struct.pack("<BBH", 0, 0, len(nodes)))
        body += struct.pack("<QQ", UNDEFINED, UNDEFINED)
        body += struct.pack("<Q", keys[0])
        for node, key in zip(nodes, keys[1:]):
            body += struct.pack("<QQ", node, key)
        full = 24 + (2 * self.INTERNAL_K + 1) * 8 + 2 * self.INTERNAL_K * 8
        body += b"\x00" * (full - len(body))
        tree = self._allocate(bytes(body))
        messages = [_message(0x11, struct.pack("<QQ", tree, heap))]
        messages += [_attribute_message(k, v) for k, v in group.attrs.items()]
        return tree, heap, self._header(messages)
