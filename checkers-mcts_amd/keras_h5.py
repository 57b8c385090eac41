"""Import of the reference's Keras model files (`neural_network.save('...h5')`,
training_pipeline.py:185-191; loaded back with load_model at :345,515-516) into
net.PolicyValueNet -- without h5py or TensorFlow.

Two parts:

  * `H5File`: a minimal pure-Python reader of the HDF5 subset h5py writes with its
    default settings (what tf.keras 2.x produces): superblock version 0/1, old-style
    groups (symbol-table message -> v1 B-tree + local heap + SNOD nodes; compact
    new-style link messages are understood too), version-1/2 object headers with
    continuation blocks, contiguous and compact dataset layouts, fixed-point / IEEE
    float / fixed-length string datatypes, attributes (string arrays such as `layer_names` /
    `weight_names`, fixed-length as h5py 2.x wrote them or variable-length through the global
    heap as h5py 3.x does).
    Chunked / filtered datasets raise: Keras does not write them for weights.
  * `load_keras_weights`: the layer mapping.  create_nn (training_pipeline.py:59-114)
    builds, in this order, Conv2D 0-6 (body), 7 (policy conv 3x3), 8 (policy conv 1x1),
    9 (value conv 1x1); BatchNormalization 0-6, 7, 8 (policy), 9 (value conv), 10 (after
    the value Dense(64)); one auto-named Dense (the value head's Dense(64)) and the named
    `policy_head` / `value_head`.  Keras numbers auto-names with a per-session counter
    (conv2d, conv2d_1, ... or conv2d_10, ... for a later model), so layers are ranked by
    their numeric suffix.  Layout conversions: Conv2D kernel (H, W, in, out) -> torch
    (out, in, H, W); Dense kernel (in, out) -> Linear weight (out, in); BatchNormalization
    gamma / beta / moving_mean / moving_variance -> weight / bias / running_mean /
    running_var (eps 1e-3 as in Keras).  Flatten is (H, W, C) on both sides
    (net.PolicyValueNet.forward), so Dense rows keep their order and the Dense(512) output
    index stays layer*64 + 8x + y (Checkers.py:433-434).
"""
import re

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5Error(ValueError):
    pass


class H5Object:
    """A group or a dataset: `kind`, `attrs`, and `links` (groups) or array metadata (datasets)."""

    def __init__(self, kind):
        self.kind = kind            # "group" | "dataset"
        self.attrs = {}
        self.links = {}             # name -> object header address
        self.shape = None
        self.dtype = None
        self.layout = None          # ("contiguous", addr, size) | ("compact", bytes)


class H5File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        self._superblock()
        self._cache = {}

    # ---- low level -------------------------------------------------------------------------
    def _u(self, off, n):
        return int.from_bytes(self.buf[off:off + n], "little")

    def _addr(self, off):
        a = self._u(off, self.so)
        return None if a == (1 << (8 * self.so)) - 1 else a + self.base

    def _superblock(self):
        b = self.buf
        off = 0
        while b[off:off + 8] != SIGNATURE:
            off = 512 if off == 0 else off * 2
            if off + 8 > len(b):
                raise H5Error("not an HDF5 file (signature not found)")
        ver = b[off + 8]
        self.base = 0
        if ver in (0, 1):
            self.so, self.sl = b[off + 13], b[off + 14]
            p = off + 24 + (4 if ver == 1 else 0)
            self.base = self._u(p, self.so)
            p += 4 * self.so                              # base, free-space info, end of file, driver info
            # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch
            self.root_addr = self._u(p + self.so, self.so) + self.base
        elif ver in (2, 3):
            self.so, self.sl = b[off + 9], b[off + 10]
            p = off + 12
            self.base = self._u(p, self.so)
            self.root_addr = self._u(p + 3 * self.so, self.so) + self.base
        else:
            raise H5Error("unsupported HDF5 superblock version %d" % ver)

    # ---- object headers --------------------------------------------------------------------
    def _messages(self, addr):
        """Yields (type, flags, payload bytes) of the object header at addr (v1 and v2 headers)."""
        b = self.buf
        if b[addr:addr + 4] == b"OHDR":
            if b[addr + 4] != 2:
                raise H5Error("unsupported object header version")
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16                                   # access, modification, change, birth times
            if flags & 0x10:
                p += 4                                    # max compact / min dense attributes
            csize = 1 << (flags & 3)
            chunk_len = self._u(p, csize)
            p += csize
            blocks = [(p, p + chunk_len)]
            order = 2 if flags & 4 else 0
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 + order <= end - 4:          # 4 bytes of checksum close every chunk
                    mtype, msize, mflags = b[p], self._u(p + 1, 2), b[p + 3]
                    body = p + 4 + order
                    if mtype == 0x10:
                        ca, cl = self._addr(body), self._u(body + self.so, self.sl)
                        if b[ca:ca + 4] != b"OCHK":
                            raise H5Error("bad object header continuation")
                        blocks.append((ca + 4, ca + cl))
                    elif mtype != 0:
                        yield mtype, mflags, b[body:body + msize]
                    p = body + msize
            return
        if b[addr] != 1:
            raise H5Error("unsupported object header version %d at %d" % (b[addr], addr))
        n_msgs = self._u(addr + 2, 2)
        size = self._u(addr + 8, 4)
        blocks = [(addr + 16, addr + 16 + size)]
        seen = 0
        while blocks and seen < n_msgs:
            p, end = blocks.pop(0)
            while p + 8 <= end and seen < n_msgs:
                mtype, msize, mflags = self._u(p, 2), self._u(p + 2, 2), b[p + 4]
                body = p + 8
                seen += 1
                if mtype == 0x10:
                    blocks.append((self._addr(body), self._addr(body) + self._u(body + self.so, self.sl)))
                elif mtype != 0:
                    yield mtype, mflags, b[body:body + msize]
                p = body + msize

    # ---- message decoders ------------------------------------------------------------------
    @staticmethod
    def _dataspace(m, sl):
        ver, rank, flags = m[0], m[1], m[2]
        p = 8 if ver == 1 else 4
        if ver == 2 and m[3] == 2:
            return None                                   # null dataspace
        return tuple(int.from_bytes(m[p + i * sl:p + (i + 1) * sl], "little") for i in range(rank))

    @staticmethod
    def _datatype(m):
        """-> (numpy dtype | ('vlen', ...) | None, size in bytes)."""
        cls, bits0 = m[0] & 15, m[1]
        size = int.from_bytes(m[4:8], "little")
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            return np.dtype("%s%s%d" % (order, "i" if bits0 & 8 else "u", size)), size
        if cls == 1:
            return np.dtype("%sf%d" % (order, size)), size
        if cls == 3:
            return np.dtype("S%d" % size), size
        if cls == 9:
            return ("vlen", bits0 & 15), size
        return None, size

    def _attribute(self, m):
        ver = m[0]
        nsz, tsz, ssz = (int.from_bytes(m[i:i + 2], "little") for i in (2, 4, 6))
        p = 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = m[p:p + nsz].split(b"\0", 1)[0].decode("utf8", "replace")
        p += pad(nsz)
        dt, esz = self._datatype(m[p:p + tsz])
        p += pad(tsz)
        shape = self._dataspace(m[p:p + ssz], self.sl)
        p += pad(ssz)
        if shape is None:
            return name, None
        count = int(np.prod(shape)) if shape else 1
        if isinstance(dt, tuple) and dt[0] == "vlen" and dt[1] == 1:      # variable-length strings (h5py >= 3 writes str / lists so)
            vals = [self._vlen_bytes(m[p + i * esz:p + (i + 1) * esz]).decode("utf8", "replace") for i in range(count)]
            return name, (np.array(vals, dtype=object).reshape(shape) if shape else vals[0])
        if not isinstance(dt, np.dtype):
            return name, None                             # other variable-length / compound types: not needed
        arr = np.frombuffer(m[p:p + count * esz], dtype=dt, count=count).reshape(shape)
        if dt.kind == "S":
            arr = np.array([s.split(b"\0", 1)[0].decode("utf8") for s in arr.reshape(-1)], dtype=object).reshape(shape)
        return name, (arr if shape else arr.reshape(-1)[0])

    def _vlen_bytes(self, ref):
        """Variable-length element {length (4), global heap collection address, object index (4)} -> bytes."""
        n = int.from_bytes(ref[0:4], "little")
        a = int.from_bytes(ref[4:4 + self.so], "little")
        idx = int.from_bytes(ref[4 + self.so:8 + self.so], "little")
        if n == 0 or a == (1 << (8 * self.so)) - 1:
            return b""
        a += self.base
        b = self.buf
        if b[a:a + 4] != b"GCOL":
            raise H5Error("bad global heap collection")
        end = a + self._u(a + 8, self.sl)
        p = a + 8 + self.sl
        while p + 8 + self.sl <= end:
            oid, osz = self._u(p, 2), self._u(p + 8, self.sl)
            if oid == 0:
                break
            if oid == idx:
                return bytes(b[p + 8 + self.sl:p + 8 + self.sl + n])
            p += 8 + self.sl + ((osz + 7) & ~7)
        raise H5Error("global heap object %d not found" % idx)

    def _layout(self, m):
        ver = m[0]
        if ver == 3:
            cls = m[1]
            if cls == 0:
                n = int.from_bytes(m[2:4], "little")
                return ("compact", bytes(m[4:4 + n]))
            if cls == 1:
                a = int.from_bytes(m[2:2 + self.so], "little")
                a = None if a == (1 << (8 * self.so)) - 1 else a + self.base
                return ("contiguous", a, int.from_bytes(m[2 + self.so:2 + self.so + self.sl], "little"))
            return ("chunked",)
        if ver in (1, 2):
            rank, cls = m[1], m[2]
            p = 8
            a = None
            if cls != 0:
                a = int.from_bytes(m[p:p + self.so], "little")
                a = None if a == (1 << (8 * self.so)) - 1 else a + self.base
                p += self.so
            p += 4 * rank
            if cls == 0:
                n = int.from_bytes(m[p:p + 4], "little")
                return ("compact", bytes(m[p + 4:p + 4 + n]))
            return ("contiguous", a, None) if cls == 1 else ("chunked",)
        raise H5Error("unsupported data layout message version %d" % ver)

    def _link(self, m):
        flags = m[1]
        p = 2
        ltype = 0
        if flags & 8:
            ltype = m[p]; p += 1
        if flags & 4:
            p += 8
        if flags & 16:
            p += 1
        lsz = 1 << (flags & 3)
        n = int.from_bytes(m[p:p + lsz], "little")
        p += lsz
        name = m[p:p + n].decode("utf8", "replace")
        p += n
        if ltype != 0:
            return name, None                             # soft / external link
        a = int.from_bytes(m[p:p + self.so], "little")
        return name, a + self.base

    # ---- groups ----------------------------------------------------------------------------
    def _heap_name(self, heap_addr, off):
        b = self.buf
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5Error("bad local heap")
        data = self._addr(heap_addr + 8 + 2 * self.sl)
        end = b.index(b"\0", data + off)
        return b[data + off:end].decode("utf8", "replace")

    def _btree_links(self, node, heap, out):
        b = self.buf
        if b[node:node + 4] == b"SNOD":
            n = self._u(node + 6, 2)
            p = node + 8
            for _ in range(n):
                out[self._heap_name(heap, self._u(p, self.so))] = self._addr(p + self.so)
                p += 2 * self.so + 24
            return
        if b[node:node + 4] != b"TREE" or b[node + 4] != 0:
            raise H5Error("bad group B-tree node")
        used = self._u(node + 6, 2)
        p = node + 8 + 2 * self.so
        for _ in range(used):
            p += self.sl                                  # key
            self._btree_links(self._addr(p), heap, out)
            p += self.so

    # ---- public ----------------------------------------------------------------------------
    def obj(self, addr):
        if addr in self._cache:
            return self._cache[addr]
        shape = dt = layout = None
        links, attrs = {}, {}
        is_group = False
        for mtype, _flags, m in self._messages(addr):
            if mtype == 0x11:
                is_group = True
                self._btree_links(self._addr_in(m, 0), self._addr_in(m, self.so), links)
            elif mtype == 0x06:
                is_group = True
                name, a = self._link(m)
                if a is not None:
                    links[name] = a
            elif mtype == 0x02:
                is_group = True
                fh = int.from_bytes(m[-2 * self.so:-self.so] if not (m[1] & 2) else m[-3 * self.so:-2 * self.so], "little")
                if fh != (1 << (8 * self.so)) - 1:
                    raise H5Error("groups with dense link storage (fractal heap) are not supported")
            elif mtype == 0x01:
                shape = self._dataspace(m, self.sl)
            elif mtype == 0x03:
                dt, _ = self._datatype(m)
            elif mtype == 0x08:
                layout = self._layout(m)
            elif mtype == 0x0B:
                raise H5Error("filtered (compressed) datasets are not supported")
            elif mtype == 0x0C:
                name, val = self._attribute(m)
                attrs[name] = val
        o = H5Object("group" if is_group or layout is None else "dataset")
        o.attrs, o.links, o.shape, o.dtype, o.layout = attrs, links, shape, dt, layout
        self._cache[addr] = o
        return o

    def _addr_in(self, m, off):
        a = int.from_bytes(m[off:off + self.so], "little")
        return a + self.base

    @property
    def root(self):
        return self.obj(self.root_addr)

    def get(self, path):
        o = self.root
        for part in [p for p in path.split("/") if p]:
            if o.kind != "group" or part not in o.links:
                raise KeyError(path)
            o = self.obj(o.links[part])
        return o

    def read(self, o):
        """Dataset -> numpy array (native byte order)."""
        if o.kind != "dataset" or not isinstance(o.dtype, np.dtype):
            raise H5Error("not a readable dataset")
        shape = o.shape or ()
        count = int(np.prod(shape)) if shape else 1
        if o.layout[0] == "compact":
            raw = o.layout[1]
        elif o.layout[0] == "contiguous":
            if o.layout[1] is None:
                return np.zeros(shape, o.dtype.newbyteorder("="))
            raw = self.buf[o.layout[1]:o.layout[1] + count * o.dtype.itemsize]
        else:
            raise H5Error("chunked datasets are not supported (Keras writes weights contiguously)")
        if len(raw) < count * o.dtype.itemsize:
            raise H5Error("truncated dataset")
        return np.frombuffer(raw, dtype=o.dtype, count=count).reshape(shape).astype(o.dtype.newbyteorder("="))

    def datasets(self, group=None, prefix=""):
        """{path relative to `group`: H5Object} of every dataset below it."""
        group = group or self.root
        out = {}
        for name, addr in group.links.items():
            o = self.obj(addr)
            path = prefix + name
            if o.kind == "group":
                out.update(self.datasets(o, path + "/"))
            else:
                out[path] = o
        return out


# ---- Keras model -> PolicyValueNet ---------------------------------------------------------------

def _rank(names, stem):
    """Layer names `stem`, `stem_1`, `stem_7`, ... in creation order (numeric suffix)."""
    pat = re.compile(r"^%s(?:_(\d+))?$" % re.escape(stem))
    found = [(int(m.group(1) or 0), n) for n in names for m in [pat.match(n)] if m]
    return [n for _, n in sorted(found)]


def read_keras_layers(path):
    """{layer name: {weight short name: float32 array}} of a Keras .h5 file (a full `model.save` file
    with its `model_weights` group, or a `save_weights` file whose root holds the layer groups)."""
    f = H5File(path)
    root = f.root
    if "model_weights" in root.links:
        root = f.obj(root.links["model_weights"])
    layers = {}
    for lname, addr in root.links.items():
        g = f.obj(addr)
        if g.kind != "group":
            continue
        ws = {}
        for wpath, d in f.datasets(g).items():
            ws[wpath.rsplit("/", 1)[-1].split(":")[0]] = f.read(d)
        layers[lname] = ws
    return layers


def keras_state_dict(layers):
    """Keras layer weights (read_keras_layers) -> state_dict of net.PolicyValueNet (numpy arrays)."""
    names = list(layers)
    convs = [n for n in _rank(names, "conv2d") if layers[n]]
    bns = [n for n in _rank(names, "batch_normalization") if layers[n]]
    denses = [n for n in _rank(names, "dense") if layers[n]]
    if len(convs) != 10 or len(bns) != 11 or len(denses) != 1 or "policy_head" not in layers or "value_head" not in layers:
        raise ValueError("not a Checkers-MCTS create_nn model: %d Conv2D, %d BatchNormalization, %d Dense layers, heads %s"
                         % (len(convs), len(bns), len(denses), [h for h in ("policy_head", "value_head") if h in layers]))
    sd = {}

    def conv(dst, lname, kshape):
        k, b = layers[lname]["kernel"], layers[lname]["bias"]
        if kshape and tuple(k.shape) != kshape:
            raise ValueError("%s: Conv2D kernel %s, expected %s" % (lname, k.shape, kshape))
        sd[dst + ".weight"] = np.ascontiguousarray(k.transpose(3, 2, 0, 1), np.float32)      # (H, W, in, out) -> (out, in, H, W)
        sd[dst + ".bias"] = np.asarray(b, np.float32)

    def bn(dst, lname):
        w = layers[lname]
        sd[dst + ".weight"], sd[dst + ".bias"] = np.asarray(w["gamma"], np.float32), np.asarray(w["beta"], np.float32)
        sd[dst + ".running_mean"] = np.asarray(w["moving_mean"], np.float32)
        sd[dst + ".running_var"] = np.asarray(w["moving_variance"], np.float32)
        sd[dst + ".num_batches_tracked"] = np.asarray(0, np.int64)

    def dense(dst, lname, kshape):
        k = layers[lname]["kernel"]
        if tuple(k.shape) != kshape:
            raise ValueError("%s: Dense kernel %s, expected %s" % (lname, k.shape, kshape))
        sd[dst + ".weight"] = np.ascontiguousarray(k.T, np.float32)                          # (in, out) -> (out, in)
        sd[dst + ".bias"] = np.asarray(layers[lname]["bias"], np.float32)

    K = int(layers[convs[0]]["kernel"].shape[3])
    for i in range(7):
        conv("body.%d.conv" % i, convs[i], (3, 3, 14 if i == 0 else K, K))
        bn("body.%d.bn" % i, bns[i])
    conv("pol1.conv", convs[7], (3, 3, K, K)); bn("pol1.bn", bns[7])
    conv("pol2.conv", convs[8], (1, 1, K, 8)); bn("pol2.bn", bns[8])
    conv("val1.conv", convs[9], (1, 1, K, 1)); bn("val1.bn", bns[9])
    dense("val_fc1", denses[0], (64, 64)); bn("val_bn", bns[10])
    dense("pol_fc", "policy_head", (512, 512))
    dense("val_fc2", "value_head", (64, 1))
    return sd, K


def num_kernels(path):
    """NUM_KERNELS of a saved Keras model (the first Conv2D's output-channel count)."""
    layers = read_keras_layers(path)
    convs = [n for n in _rank(list(layers), "conv2d") if layers[n]]
    if not convs:
        raise ValueError("no Conv2D layer in %s" % path)
    return int(layers[convs[0]]["kernel"].shape[3])


def load_keras_weights(path):
    """Keras .h5 model file -> net.PolicyValueNet (float32, eval mode, on the CPU)."""
    import torch
    from .net import PolicyValueNet
    sd, K = keras_state_dict(read_keras_layers(path))
    net = PolicyValueNet(K)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return net.eval()


# ---- PolicyValueNet -> Keras model file --------------------------------------------------------------
#
# The other direction: `neural_network.save('...h5')` (training_pipeline.py:185-191, and ModelCheckpoint in train_nn,
# :139-145) -- a file the reference's `load_model` (training_pipeline.py:345,515-516; train_Checkers.py:164;
# play_Checkers.py:109) can open, written without h5py or TensorFlow.  `H5Writer` emits the same HDF5 subset the reader above
# understands and that h5py 2.10 / libhdf5 1.10 produce with default settings: superblock version 0, version-1 object
# headers, old-style groups (symbol-table message, v1 B-tree, local heap, one symbol node per group), contiguous
# little-endian datasets, fixed-length string attributes (what h5py makes of the `bytes` tf.keras 2.2 stores:
# hdf5_format.save_model_to_hdf5 encodes model_config / training_config / layer_names / weight_names to utf-8 bytes, and
# its loader calls .decode() on what it reads back).  Checked against a real HDF5 library in the build container
# (tests/test_keras_h5_cpu.py reads the written file back with /opt/conda's h5py when that interpreter is present).

UNDEF = (1 << 64) - 1
_LEAF_K, _INTERNAL_K = 32, 16          # symbol-node / B-tree fan-out parameters, recorded in the superblock


def _pad8(b):
    return bytes(b) + b"\0" * (-len(b) % 8)


class _Node:
    def __init__(self, name, data=None):
        self.name, self.data, self.attrs, self.children = name, data, [], []      # data is None: a group


class H5Writer:
    """Build a tree with group() / dataset() / attr(), then write(path)."""

    def __init__(self):
        self.root = _Node("/")

    def group(self, parent, name):
        g = _Node(name)
        parent.children.append(g)
        return g

    def dataset(self, parent, name, array):
        """`name` may contain '/' (h5py creates the intermediate groups, as in Keras' '<layer>/kernel:0')."""
        parts = name.split("/")
        for p in parts[:-1]:
            nxt = next((c for c in parent.children if c.name == p and c.data is None), None)
            parent = nxt or self.group(parent, p)
        a = np.asarray(array)
        if a.dtype not in (np.dtype("<f4"), np.dtype("<f8"), np.dtype("<i8")):
            raise H5Error("H5Writer: unsupported dataset type %s" % a.dtype)
        d = _Node(parts[-1], np.ascontiguousarray(a))
        parent.children.append(d)
        return d

    @staticmethod
    def attr(node, name, value):
        """value: bytes / str (fixed-length string scalar), a list of bytes / str (1-D fixed-length string array; an empty
        list becomes an empty float64 array, which is what h5py makes of []), or a numpy scalar / array of f4, f8, i8."""
        node.attrs.append((name, value))

    # ---- encoders --------------------------------------------------------------------------------
    @staticmethod
    def _dtype_msg(dt, strlen=0):
        if dt == "S":                                       # fixed-length string, null-padded, ASCII (numpy 'S<n>')
            return bytes([0x13, 0x01, 0, 0]) + int(strlen).to_bytes(4, "little")
        if dt == np.dtype("<f4"):
            return bytes([0x11, 0x20, 31, 0]) + (4).to_bytes(4, "little") + bytes([0, 0, 32, 0, 23, 8, 0, 23]) + (127).to_bytes(4, "little")
        if dt == np.dtype("<f8"):
            return bytes([0x11, 0x20, 63, 0]) + (8).to_bytes(4, "little") + bytes([0, 0, 64, 0, 52, 11, 0, 52]) + (1023).to_bytes(4, "little")
        if dt == np.dtype("<i8"):
            return bytes([0x10, 0x08, 0, 0]) + (8).to_bytes(4, "little") + bytes([0, 0, 64, 0])
        raise H5Error("H5Writer: unsupported type %r" % (dt,))

    @staticmethod
    def _space_msg(shape):
        if shape is None:                                   # scalar
            return bytes([1, 0, 0, 0, 0, 0, 0, 0])
        return bytes([1, len(shape), 0, 0, 0, 0, 0, 0]) + b"".join(int(d).to_bytes(8, "little") for d in shape)

    def _attr_msg(self, name, value):
        if isinstance(value, str):
            value = value.encode("utf8")
        if isinstance(value, (bytes, bytearray)):
            dt, sp, raw = self._dtype_msg("S", max(1, len(value))), self._space_msg(None), bytes(value) or b"\0"
        elif isinstance(value, (list, tuple)):
            items = [v.encode("utf8") if isinstance(v, str) else bytes(v) for v in value]
            if not items:
                dt, sp, raw = self._dtype_msg(np.dtype("<f8")), self._space_msg((0,)), b""
            else:
                n = max(1, max(len(v) for v in items))
                dt, sp, raw = self._dtype_msg("S", n), self._space_msg((len(items),)), b"".join(v.ljust(n, b"\0") for v in items)
        else:
            a = np.asarray(value)
            dt, sp, raw = self._dtype_msg(a.dtype), self._space_msg(a.shape if a.shape else None), np.ascontiguousarray(a).tobytes()
        nm = name.encode("utf8") + b"\0"
        body = bytes([1, 0]) + len(nm).to_bytes(2, "little") + len(dt).to_bytes(2, "little") + len(sp).to_bytes(2, "little")
        body += _pad8(nm) + _pad8(dt) + _pad8(sp) + raw
        if len(body) > 65528:
            raise H5Error("attribute %r is %d bytes: above the 64-KB limit of an object header message (the same limit "
                          "h5py / Keras hit)" % (name, len(body)))
        return 0x0C, body

    def _header(self, msgs):
        """Version-1 object header holding all messages in its first chunk."""
        body = b""
        for mtype, data in msgs:
            data = _pad8(data)
            body += int(mtype).to_bytes(2, "little") + len(data).to_bytes(2, "little") + bytes([0, 0, 0, 0]) + data
        return bytes([1, 0]) + len(msgs).to_bytes(2, "little") + (1).to_bytes(4, "little") + len(body).to_bytes(4, "little") + b"\0" * 4 + body

    def _put(self, blob):
        self.buf += b"\0" * (-len(self.buf) % 8)
        at = len(self.buf)
        self.buf += blob
        return at

    def _write_node(self, node):
        """Returns (object header address, B-tree address, local heap address) -- the last two 0 for a dataset."""
        attrs = [self._attr_msg(n, v) for n, v in node.attrs]
        if node.data is not None:
            a = node.data
            raw_at = self._put(a.tobytes()) if a.size else UNDEF
            layout = bytes([3, 1]) + int(raw_at).to_bytes(8, "little") + int(a.nbytes).to_bytes(8, "little")
            msgs = [(0x01, self._space_msg(a.shape)), (0x03, self._dtype_msg(a.dtype)), (0x05, bytes([2, 1, 0, 0])), (0x08, layout)]
            return self._put(self._header(msgs + attrs)), 0, 0
        if len(node.children) > 2 * _LEAF_K:
            raise H5Error("H5Writer: more than %d links in one group" % (2 * _LEAF_K))
        kids = sorted(node.children, key=lambda c: c.name.encode("utf8"))
        if len({c.name for c in kids}) != len(kids):
            raise H5Error("H5Writer: duplicate link name in group %r" % node.name)
        placed = [(c, self._write_node(c)) for c in kids]
        # local heap: the empty name at offset 0, then the link names
        heap_data, offs = bytearray(8), []
        for c in kids:
            offs.append(len(heap_data))
            heap_data += _pad8(c.name.encode("utf8") + b"\0")
        data_at = self._put(bytes(heap_data))
        heap_at = self._put(b"HEAP" + bytes(4) + len(heap_data).to_bytes(8, "little") + (1).to_bytes(8, "little") + data_at.to_bytes(8, "little"))   # free list: 1 = none (H5HL_FREE_NULL)
        snod = bytearray(b"SNOD" + bytes([1, 0]) + len(kids).to_bytes(2, "little"))
        for (c, (hdr, bt, hp)), off in zip(placed, offs):
            snod += off.to_bytes(8, "little") + hdr.to_bytes(8, "little")
            snod += ((1).to_bytes(4, "little") + bytes(4) + bt.to_bytes(8, "little") + hp.to_bytes(8, "little")) if c.data is None else bytes(24)
        snod += bytes(8 + 2 * _LEAF_K * 40 - len(snod))
        snod_at = self._put(bytes(snod))
        tree = bytearray(b"TREE" + bytes([0, 0]) + (1 if kids else 0).to_bytes(2, "little") + UNDEF.to_bytes(8, "little") * 2)
        tree += (0).to_bytes(8, "little")
        if kids:
            tree += snod_at.to_bytes(8, "little") + offs[-1].to_bytes(8, "little")
        tree += bytes(24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8 - len(tree))
        tree_at = self._put(bytes(tree))
        stab = tree_at.to_bytes(8, "little") + heap_at.to_bytes(8, "little")
        return self._put(self._header([(0x11, stab)] + attrs)), tree_at, heap_at

    def write(self, path):
        self.buf = bytearray(96)                            # the superblock is filled in last
        hdr, tree, heap = self._write_node(self.root)
        self.buf += b"\0" * (-len(self.buf) % 8)
        sb = SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + _LEAF_K.to_bytes(2, "little") + _INTERNAL_K.to_bytes(2, "little") + bytes(4)
        sb += (0).to_bytes(8, "little") + UNDEF.to_bytes(8, "little") + len(self.buf).to_bytes(8, "little") + UNDEF.to_bytes(8, "little")
        sb += (0).to_bytes(8, "little") + hdr.to_bytes(8, "little") + (1).to_bytes(4, "little") + bytes(4) + tree.to_bytes(8, "little") + heap.to_bytes(8, "little")
        assert len(sb) == 96
        self.buf[0:96] = sb
        with open(path, "wb") as f:
            f.write(self.buf)


def _f32(x):
    """A Python float that went through float32, as Keras' get_config() reports regularisation factors and Adam's settings."""
    return float(np.float32(x))


def keras_layer_table(num_kernels):
    """create_nn's layers (training_pipeline.py:59-114) in tf.keras' `model.layers` order, each as (name, kind, detail,
    inbound layer, torch module prefix).  Names: the per-class counters of a fresh Keras session in creation order.
    Order: tf.keras sorts a functional model's layers by decreasing depth (longest path to an output) and, inside one
    depth, by the order in which a depth-first walk from the outputs [policy_head, value_head] first meets them
    (network._map_graph_network) -- the order in which save_weights lists `layer_names` and load_weights pairs them."""
    K = int(num_kernels)
    t = [("input_1", "input", None, None, None)]
    prev = "input_1"
    for i in range(7):
        c, b = "conv2d" + ("_%d" % i if i else ""), "batch_normalization" + ("_%d" % i if i else "")
        t += [(c, "conv", (K, 3), prev, "body.%d.conv" % i), (b, "bn", 3, c, "body.%d.bn" % i)]
        prev = b
    t += [("conv2d_7", "conv", (K, 3), prev, "pol1.conv"), ("conv2d_9", "conv", (1, 1), prev, "val1.conv"),
          ("batch_normalization_7", "bn", 3, "conv2d_7", "pol1.bn"), ("batch_normalization_9", "bn", 3, "conv2d_9", "val1.bn"),
          ("conv2d_8", "conv", (8, 1), "batch_normalization_7", "pol2.conv"), ("flatten_1", "flatten", None, "batch_normalization_9", None),
          ("batch_normalization_8", "bn", 3, "conv2d_8", "pol2.bn"), ("dense", "dense", (64, "relu"), "flatten_1", "val_fc1"),
          ("flatten", "flatten", None, "batch_normalization_8", None), ("batch_normalization_10", "bn", 1, "dense", "val_bn"),
          ("policy_head", "dense", (512, "softmax"), "flatten", "pol_fc"), ("value_head", "dense", (1, "tanh"), "batch_normalization_10", "val_fc2")]
    return t


def keras_model_config(num_kernels=128, conv_reg=0.001, dense_reg=0.001):
    """The `model_config` attribute tf.keras 2.2 (`keras_version` 2.3.0-tf) stores for create_nn's model: what load_model
    rebuilds the architecture from.  Restated from tf.keras' layer get_config() methods (TensorFlow is absent here)."""
    def reg(v):
        return {"class_name": "L1L2", "config": {"l1": 0.0, "l2": _f32(v)}}
    zeros, ones = {"class_name": "Zeros", "config": {}}, {"class_name": "Ones", "config": {}}
    glorot = {"class_name": "GlorotUniform", "config": {"seed": None}}
    layers = []
    for name, kind, detail, inbound, _ in keras_layer_table(num_kernels):
        if kind == "input":
            cfg, cls = {"batch_input_shape": [None, 8, 8, 14], "dtype": "float32", "sparse": False, "ragged": False, "name": name}, "InputLayer"
        elif kind == "conv":
            cfg, cls = {"name": name, "trainable": True, "dtype": "float32", "filters": detail[0], "kernel_size": [detail[1]] * 2,
                        "strides": [1, 1], "padding": "same", "data_format": "channels_last", "dilation_rate": [1, 1],
                        "activation": "relu", "use_bias": True, "kernel_initializer": glorot, "bias_initializer": zeros,
                        "kernel_regularizer": reg(conv_reg), "bias_regularizer": reg(conv_reg), "activity_regularizer": None,
                        "kernel_constraint": None, "bias_constraint": None}, "Conv2D"
        elif kind == "bn":
            cfg, cls = {"name": name, "trainable": True, "dtype": "float32", "axis": [detail], "momentum": 0.99, "epsilon": 0.001,
                        "center": True, "scale": True, "beta_initializer": zeros, "gamma_initializer": ones,
                        "moving_mean_initializer": zeros, "moving_variance_initializer": ones, "beta_regularizer": None,
                        "gamma_regularizer": None, "beta_constraint": None, "gamma_constraint": None}, "BatchNormalization"
        elif kind == "flatten":
            cfg, cls = {"name": name, "trainable": True, "dtype": "float32", "data_format": "channels_last"}, "Flatten"
        else:
            cfg, cls = {"name": name, "trainable": True, "dtype": "float32", "units": detail[0], "activation": detail[1], "use_bias": True,
                        "kernel_initializer": glorot, "bias_initializer": zeros, "kernel_regularizer": reg(dense_reg),
                        "bias_regularizer": reg(dense_reg), "activity_regularizer": None, "kernel_constraint": None,
                        "bias_constraint": None}, "Dense"
        layers.append({"class_name": cls, "config": cfg, "name": name, "inbound_nodes": [[[inbound, 0, 0, {}]]] if inbound else []})
    return {"class_name": "Model", "config": {"name": "model", "layers": layers, "input_layers": [["input_1", 0, 0]],
                                              "output_layers": [["policy_head", 0, 0], ["value_head", 0, 0]]}}


def keras_training_config(policy_loss_weight=1.0, value_loss_weight=1.0):
    """The `training_config` attribute of model.compile(...) in create_nn (training_pipeline.py:108-113): losses per head,
    loss weights, Adam() with tf.keras' defaults -- what load_model compiles the rebuilt model with."""
    return {"loss": {"policy_head": "categorical_crossentropy", "value_head": "mse"}, "metrics": None, "weighted_metrics": None,
            "loss_weights": {"policy_head": policy_loss_weight, "value_head": value_loss_weight}, "sample_weight_mode": None,
            "optimizer_config": {"class_name": "Adam", "config": {"name": "Adam", "learning_rate": _f32(0.001), "decay": 0.0,
                                                                  "beta_1": _f32(0.9), "beta_2": _f32(0.999), "epsilon": 1e-07,
                                                                  "amsgrad": False}}}


def keras_layer_weights(state_dict):
    """state_dict of net.PolicyValueNet -> [(layer name, [(weight name, float32 array)])] in Keras layout and `model.layers`
    order: the inverse of keras_state_dict (Conv2D (out,in,H,W) -> (H,W,in,out); Linear (out,in) -> (in,out))."""
    sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in state_dict.items()}
    K = int(sd["body.0.conv.weight"].shape[0])
    out = []
    for name, kind, _, _, prefix in keras_layer_table(K):
        if kind == "conv":
            ws = [("kernel:0", sd[prefix + ".weight"].transpose(2, 3, 1, 0)), ("bias:0", sd[prefix + ".bias"])]
        elif kind == "dense":
            ws = [("kernel:0", sd[prefix + ".weight"].T), ("bias:0", sd[prefix + ".bias"])]
        elif kind == "bn":
            ws = [("gamma:0", sd[prefix + ".weight"]), ("beta:0", sd[prefix + ".bias"]),
                  ("moving_mean:0", sd[prefix + ".running_mean"]), ("moving_variance:0", sd[prefix + ".running_var"])]
        else:
            ws = []
        out.append((name, [(name + "/" + w, np.ascontiguousarray(a, dtype="<f4")) for w, a in ws]))
    return out


def save_keras_model(net, path, conv_reg=None, dense_reg=None, policy_loss_weight=None, value_loss_weight=None):
    """`neural_network.save(path)` for a net.PolicyValueNet: a tf.keras 2.2 HDF5 model file (model_config,
    training_config, /model_weights) for the reference's load_model.  Regularisation factors and loss weights default to
    the attributes train.create_nn / train_nn put on the module (create_nn kwargs, training_pipeline.py:56-60)."""
    import json
    pick = lambda v, attr, dflt: float(v if v is not None else getattr(net, attr, dflt))
    layers = keras_layer_weights(net.state_dict())
    K = int(net.state_dict()["body.0.conv.weight"].shape[0])
    w = H5Writer()
    H5Writer.attr(w.root, "keras_version", b"2.3.0-tf")
    H5Writer.attr(w.root, "backend", b"tensorflow")
    H5Writer.attr(w.root, "model_config", json.dumps(keras_model_config(K, pick(conv_reg, "conv_reg", 0.001), pick(dense_reg, "dense_reg", 0.001))).encode("utf8"))
    H5Writer.attr(w.root, "training_config", json.dumps(keras_training_config(pick(policy_loss_weight, "policy_loss_weight", 1.0),
                                                                              pick(value_loss_weight, "value_loss_weight", 1.0))).encode("utf8"))
    g = w.group(w.root, "model_weights")
    H5Writer.attr(g, "layer_names", [n.encode("utf8") for n, _ in layers])
    H5Writer.attr(g, "backend", b"tensorflow")
    H5Writer.attr(g, "keras_version", b"2.3.0-tf")
    for name, weights in layers:
        lg = w.group(g, name)
        H5Writer.attr(lg, "weight_names", [wn.encode("utf8") for wn, _ in weights])
        for wn, a in weights:
            w.dataset(lg, wn, a)
    w.write(path)
    return path
