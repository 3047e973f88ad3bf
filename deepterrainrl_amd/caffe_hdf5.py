"""Caffe HDF5 model files without an HDF5 library: just enough of the HDF5 1.8 file format to read (and write) what
caffe::Net::ToHDF5 / CopyTrainedLayersFromHDF5 exchange -- the `.h5` policies the reference saves and loads
(cNeuralNet::LoadModel / OutputModel, learning/NeuralNet.cpp:81-108, 110-135).

File layout handled (HDF5 File Format Specification v1/v2 subset, what libhdf5 1.8 / "earliest" writes):
  superblock v0 -> root symbol-table entry -> object header v1 -> symbol-table message -> v1 B-tree ("TREE", node type 0) +
  local heap ("HEAP") -> symbol-table nodes ("SNOD") -> links; datasets: object header v1 with dataspace (0x0001),
  datatype (0x0003; IEEE little-endian float32 / float64) and data-layout (0x0008, version 3, contiguous or compact) messages,
  header continuation blocks (0x0010). Chunked / filtered datasets, new-style groups and other superblock versions are rejected.
Caffe's layout: /data/<layer name>/<blob index> (float32, blob shape); layers without parameters are empty groups.

The reader is pinned by tests/golden/caffe_model_small.h5, a file produced by the real HDF5 library (h5py, see
tests/golden/make_hdf5_fixture.py); the writer is checked against the reader and, where h5py is available, against libhdf5.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(RuntimeError):
    pass


class _Reader:
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != b"\x89HDF\r\n\x1a\n":
            raise H5Error("not an HDF5 file (signature)")
        ver = buf[8]
        if ver != 0:
            raise H5Error("unsupported superblock version %d (only v0, as written by HDF5 1.8 / Caffe)" % ver)
        self.so, self.sl = buf[13], buf[14]
        if (self.so, self.sl) != (8, 8):
            raise H5Error("unsupported offset/length sizes")
        self.base = struct.unpack_from("<Q", buf, 24)[0]
        # root group symbol table entry follows the four addresses (base, free-space, eof, driver)
        self.root = self._ste(24 + 4 * 8)

    def _ste(self, off):
        name_off, ohdr, cache = struct.unpack_from("<QQI", self.b, off)
        scratch = self.b[off + 24:off + 40]
        return {"name_off": name_off, "ohdr": ohdr, "cache": cache, "btree": struct.unpack_from("<Q", scratch, 0)[0], "heap": struct.unpack_from("<Q", scratch, 8)[0]}

    def _messages(self, addr):
        """Object header v1 -> list of (type, bytes)."""
        b = self.b
        ver, _, nmsg, _refs, hsize = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise H5Error("unsupported object header version %d" % ver)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, pos)
                body = b[pos + 8:pos + 8 + msize]
                if mtype == 0x0010:
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((caddr, clen))
                out.append((mtype, body))
                pos += 8 + msize
        return out

    def _heap_name(self, heap_addr, off):
        if self.b[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5Error("bad local heap signature")
        data_addr = struct.unpack_from("<Q", self.b, heap_addr + 24)[0]
        p = data_addr + off
        return self.b[p:self.b.index(b"\0", p)].decode()

    def _btree_entries(self, addr, heap):
        b = self.b
        if addr == UNDEF:
            return
        sig = b[addr:addr + 4]
        if sig == b"TREE":
            ntype, level, used = struct.unpack_from("<BBH", b, addr + 4)
            if ntype != 0:
                raise H5Error("unexpected B-tree node type")
            p = addr + 24
            for i in range(used):
                child = struct.unpack_from("<Q", b, p + 8)[0]   # key_i, child_i, key_i+1 ...
                for e in self._btree_entries(child, heap):
                    yield e
                p += 16
        elif sig == b"SNOD":
            n = struct.unpack_from("<H", b, addr + 6)[0]
            for i in range(n):
                e = self._ste(addr + 8 + 40 * i)
                yield self._heap_name(heap, e["name_off"]), e["ohdr"]
        else:
            raise H5Error("bad group node signature %r" % sig)

    def links(self, ohdr):
        """Child name -> object header address of a (old-style) group; None if the object is not a group."""
        for mtype, body in self._messages(ohdr):
            if mtype == 0x0011:
                bt, heap = struct.unpack_from("<QQ", body, 0)
                return dict(self._btree_entries(bt, heap))
        return None

    def dataset(self, ohdr):
        shape = dtype = None
        layout = None
        for mtype, body in self._messages(ohdr):
            if mtype == 0x0001:
                ver, rank, flags = struct.unpack_from("<BBB", body, 0)
                off = 8 if ver == 1 else 4
                shape = struct.unpack_from("<%dQ" % rank, body, off) if rank else ()
            elif mtype == 0x0003:
                cls = body[0] & 0x0F
                size = struct.unpack_from("<I", body, 4)[0]
                if cls != 1 or (body[1] & 1) != 0 or size not in (4, 8):
                    raise H5Error("only little-endian IEEE float32 / float64 datasets are supported")
                dtype = np.float32 if size == 4 else np.float64
            elif mtype == 0x0008:
                ver, lclass = body[0], body[1]
                if ver != 3:
                    raise H5Error("unsupported data layout message version %d" % ver)
                if lclass == 1:
                    layout = ("contiguous",) + struct.unpack_from("<QQ", body, 2)
                elif lclass == 0:
                    n = struct.unpack_from("<H", body, 2)[0]
                    layout = ("compact", bytes(body[4:4 + n]))
                else:
                    raise H5Error("chunked datasets are not supported (Caffe writes contiguous blobs)")
        if shape is None or dtype is None or layout is None:
            raise H5Error("object is not a simple dataset")
        n = int(np.prod(shape)) if len(shape) else 1
        if layout[0] == "contiguous":
            addr, size = layout[1], layout[2]
            if addr == UNDEF:
                return np.zeros(shape, dtype)
            arr = np.frombuffer(self.b, dtype=dtype, count=n, offset=addr + self.base)
        else:
            arr = np.frombuffer(layout[1], dtype=dtype, count=n)
        return arr.reshape(shape).copy()


def read_caffe_model(path):
    """{layer name: [blob0, blob1, ...]} from /data/<layer>/<i> (layers without blobs are omitted)."""
    r = _Reader(open(path, "rb").read())
    top = r.links(r.root["ohdr"])
    if top is None or "data" not in top:
        raise H5Error("no /data group: not a Caffe HDF5 model")
    out = {}
    for layer, addr in (r.links(top["data"]) or {}).items():
        blobs = r.links(addr) or {}
        idx = sorted((int(k), a) for k, a in blobs.items() if k.isdigit())
        if idx:
            out[layer] = [r.dataset(a) for _, a in idx]
    return out


MACE_LAYERS = ("terr_conv0", "terr_conv1", "terr_conv2", "terr_ip0", "ip0", "val_ip0", "val_ip1")


def mace_layer_names(n_frags):
    names = list(MACE_LAYERS)
    for f in range(n_frags):
        names += ["a%d_ip0" % f, "a%d_ip1" % f]
    return names


def load_mace_weights(path, n_frags=3):
    """Flat float32 weight vector in the rollout engine's blob order (dtrl_set_policy) from a Caffe .h5 model of the MACE family,
    matched by layer name as Net::CopyTrainedLayersFromHDF5 does."""
    blobs = read_caffe_model(path)
    parts = []
    for name in mace_layer_names(n_frags):
        if name not in blobs or len(blobs[name]) < 2:
            raise H5Error("layer %s (weights + bias) missing from %s" % (name, path))
        parts += [np.asarray(blobs[name][0], np.float32).reshape(-1), np.asarray(blobs[name][1], np.float32).reshape(-1)]
    return np.concatenate(parts)


# ------------------------------------------------------------------------------------------------------------------
# writer (HDF5 1.8 layout: superblock v0, old-style groups, object headers v1, contiguous float32 datasets)

class _Writer:
    def __init__(self):
        self.buf = bytearray(96)   # superblock v0 (56 bytes + root symbol table entry 40 bytes), filled in at the end

    def _align(self):
        while len(self.buf) % 8:
            self.buf += b"\0"

    def _alloc(self, data):
        self._align()
        addr = len(self.buf)
        self.buf += data
        return addr

    @staticmethod
    def _msg(mtype, body, flags=0):
        body = bytes(body) + b"\0" * (-len(body) % 8)
        return struct.pack("<HHB3x", mtype, len(body), flags) + body

    def _ohdr(self, msgs):
        payload = b"".join(msgs)
        return self._alloc(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(payload)) + payload)

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        data_addr = self._alloc(arr.tobytes()) if arr.size else UNDEF
        rank = arr.ndim
        space = struct.pack("<BBB5x", 1, rank, 0) + struct.pack("<%dQ" % rank, *arr.shape)
        # IEEE float32 LE: class 1 v1, bit field (byte order LE, pad 0, mantissa norm 2 = implied msb, sign location 31), size 4,
        # properties: bit offset 0, precision 32, exp loc 23, exp size 8, mant loc 0, mant size 23, exp bias 127
        dtype = struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        fill = struct.pack("<BBBB", 2, 2, 2, 0)            # fill value message v2: alloc late, write if set, undefined
        layout = struct.pack("<BBQQ", 3, 1, data_addr, arr.nbytes)
        return self._ohdr([self._msg(0x0001, space), self._msg(0x0003, dtype, 1), self._msg(0x0005, fill), self._msg(0x0008, layout)])

    def group(self, children):
        """children: {name: object header address}. Returns (ohdr, btree, heap)."""
        names = sorted(children)
        heap_data = bytearray(b"\0" * 8)                   # offset 0: empty string (the B-tree's lowest key)
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += n.encode() + b"\0"
            heap_data += b"\0" * (-len(heap_data) % 8)
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)             # free block: next = 1 (end of list), size 16
        heap_data_addr = self._alloc(bytes(heap_data))
        heap_addr = self._alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, heap_data_addr))
        snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
        for n in names:
            snod += struct.pack("<QQI4x16x", offs[n], children[n], 0)
        leaf_k = 4
        snod += b"\0" * (40 * (2 * leaf_k - len(names)))
        if len(names) > 2 * leaf_k:
            raise H5Error("writer supports at most %d links per group node" % (2 * leaf_k))
        snod_addr = self._alloc(snod)
        internal_k = 16
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, offs[names[-1]] if names else 0)
        tree += b"\0" * (16 * (2 * internal_k - 1))
        btree_addr = self._alloc(tree)
        ohdr = self._ohdr([self._msg(0x0011, struct.pack("<QQ", btree_addr, heap_addr))])
        return ohdr, btree_addr, heap_addr

    def finish(self, root):
        ohdr, btree, heap = root
        self._align()
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQI4xQQ", 0, ohdr, 1, btree, heap)
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_caffe_model(path, layers):
    """layers: {layer name: [blob arrays]} -> HDF5 file in Caffe's /data/<layer>/<i> layout (at most 8 blobs per layer; layer groups are
    split over several symbol-table nodes automatically by nesting is NOT done: at most 8 layers per file node -> the MACE nets' 13
    parameter layers are written through a two-level B-tree)."""
    w = _Writer()
    layer_nodes = {name: w.group({str(i): w.dataset(b) for i, b in enumerate(blobs)}) [0] for name, blobs in layers.items()}
    data = _big_group(w, layer_nodes)
    root = w.group({"data": data[0]})
    open(path, "wb").write(w.finish(root))


def _big_group(w, children):
    """Group with any number of links: one SNOD per <= 8 names under a single level-0 B-tree node (up to 32 SNODs)."""
    names = sorted(children)
    if len(names) <= 8:
        return w.group(children)
    heap_data = bytearray(b"\0" * 8)
    offs = {}
    for n in names:
        offs[n] = len(heap_data)
        heap_data += n.encode() + b"\0"
        heap_data += b"\0" * (-len(heap_data) % 8)
    free_off = len(heap_data)
    heap_data += struct.pack("<QQ", 1, 16)
    heap_data_addr = w._alloc(bytes(heap_data))
    heap_addr = w._alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, heap_data_addr))
    chunks = [names[i:i + 8] for i in range(0, len(names), 8)]
    if len(chunks) > 32:
        raise H5Error("too many links for the minimal writer")
    snods = []
    for ch in chunks:
        s = b"SNOD" + struct.pack("<BBH", 1, 0, len(ch))
        for n in ch:
            s += struct.pack("<QQI4x16x", offs[n], children[n], 0)
        s += b"\0" * (40 * (8 - len(ch)))
        snods.append(w._alloc(s))
    tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(chunks), UNDEF, UNDEF) + struct.pack("<Q", 0)
    for ch, addr in zip(chunks, snods):
        tree += struct.pack("<QQ", addr, offs[ch[-1]])
    tree += b"\0" * (24 + 8 + 16 * 32 - len(tree))
    btree_addr = w._alloc(tree)
    ohdr = w._ohdr([w._msg(0x0011, struct.pack("<QQ", btree_addr, heap_addr))])
    return ohdr, btree_addr, heap_addr
