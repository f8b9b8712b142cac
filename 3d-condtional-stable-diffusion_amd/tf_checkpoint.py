"""TensorFlow checkpoint interchange without TensorFlow.

The reference saves and restores its models with ``model.save_weights(prefix)`` / ``model.load_weights(prefix)`` in the TF2
object-based checkpoint format (main_conditional_dm.py:156-160, 174-183, 207-213; ModelCheckpoint(save_weights_only=True),
main_rnsvqvae.py:223-227).  TensorFlow is not installed here and no checkpoint ships with the reference, so this module is
written from the published formats and validated only against files produced by its own writer (tests/test_tf_checkpoint.py):

* ``<prefix>.index`` — a LevelDB-style sorted string table (data blocks of prefix-compressed entries + restart array, block
  trailer = compression byte + masked CRC32C, index block, 48-byte footer with magic 0xdb4775248b80fb57).  Key "" holds a
  BundleHeaderProto, every other key a BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}.
* ``<prefix>.data-0000i-of-0000n`` — the raw little-endian tensor bytes.
* key ``_CHECKPOINTABLE_OBJECT_GRAPH`` — a string tensor with the serialized TrackableObjectGraph: per object its children
  (node_id, local_name) and its variables (name, full_name, checkpoint_key).

Mapping a Keras checkpoint to this package's flat names does NOT rely on the ``layer_with_weights-N`` indices (Keras numbers them
by graph depth, which depends on every op-lambda layer of the functional graph); it uses the variables' ``full_name``
("conv3d_12/kernel"): Keras numbers auto-named layers per class in creation order, and creation order is the order of
``build_model`` (conditional_dm3d.py:348-414), i.e. of ``weights.walk``.  Inside a CrossAttentionBlock / AttentionBlock the
attribute names of the reference class (norm, norm1..3, proj_in, proj_out, query, key, value, proj) are used directly.
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

MAGIC = 0xDB4775248B80FB57
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
VALUE_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_STRING, DT_INT64, DT_BOOL, DT_HALF = 1, 2, 3, 7, 9, 10, 19
_NP = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_INT64: np.int64, DT_BOOL: np.bool_, DT_HALF: np.float16}
_DT = {np.dtype(v): k for k, v in _NP.items()}


# ---- CRC32C (Castagnoli), masked as LevelDB / TensorFlow store it ------------------------------------------------------------
def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC = _crc_table()


def crc32c(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _CRC[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire format (just enough) --------------------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, pos: int) -> Tuple[int, int]:
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def parse_message(buf) -> Dict[int, list]:
    """field number -> list of raw values (ints for varint / fixed, bytes for length-delimited)."""
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")
        out.setdefault(field, []).append(v)
    return out


def _field(num: int, wire: int, payload: bytes) -> bytes:
    return _varint((num << 3) | wire) + payload


def _f_varint(num, v): return _field(num, 0, _varint(v))
def _f_bytes(num, b): return _field(num, 2, _varint(len(b)) + b)


def _sint64(v: int) -> int:
    """a protobuf int64 read as a varint: two's complement for negative values"""
    return v - (1 << 64) if v >= (1 << 63) else v


# ---- OrderedCode (tensorflow/core/lib/strings/ordered_code.cc), as far as slice keys need it -----------------------------------
def _oc_num_increasing(v: int) -> bytes:
    body = v.to_bytes((v.bit_length() + 7) // 8, "big") if v else b""
    return bytes([len(body)]) + body


def _oc_string(b: bytes) -> bytes:
    return b"".join(b"\x00\xff" if c == 0 else (b"\xff\x00" if c == 255 else bytes([c])) for c in b) + b"\x00\x01"


def _oc_signed_increasing(val: int) -> bytes:
    x = ~val if val < 0 else val
    if x < 64:
        return bytes([(0x80 ^ val) & 0xFF])
    n = 2
    while 7 * n - 1 < x.bit_length():                           # n bytes carry 7 n - 1 value bits behind the unary length header
        n += 1
    buf = bytearray((val & ((1 << 80) - 1)).to_bytes(10, "big"))
    header = [(0, 0), (0x80, 0), (0xC0, 0), (0xE0, 0), (0xF0, 0), (0xF8, 0), (0xFC, 0), (0xFE, 0), (0xFF, 0), (0xFF, 0x80), (0xFF, 0xC0)][n]
    buf[10 - n] ^= header[0]
    buf[11 - n] ^= header[1]
    return bytes(buf[10 - n:])


def encode_slice_key(name: str, extents) -> bytes:
    """checkpoint::EncodeTensorNameSlice: 0, the name, the rank, then (start, length) per axis with (-1, -1) for a whole axis."""
    out = _oc_num_increasing(0) + _oc_string(name.encode()) + _oc_num_increasing(len(extents))
    for st, ln in extents:
        if ln < 0:
            st = -1
        out += _oc_signed_increasing(st) + _oc_signed_increasing(ln)
    return out


# ---- sorted string table -----------------------------------------------------------------------------------------------------
def _read_block(f, offset: int, size: int, verify: bool) -> bytes:
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError("truncated table block")
    body, ctype, crc = raw[:size], raw[size], struct.unpack("<I", raw[size + 1:])[0]
    if ctype != 0:
        raise ValueError("compressed table blocks are not supported (TensorFlow writes bundle indices uncompressed)")
    if verify and mask_crc(crc32c(raw[:size + 1])) != crc:
        raise ValueError("table block checksum mismatch")
    return body


def _block_entries(block: bytes):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 * (n_restarts + 1)
    pos, key = 0, b""
    while pos < end:
        shared, pos = _read_varint(block, pos)
        unshared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path: str, verify: bool = True) -> Dict[bytes, bytes]:
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if size < 48:
            raise ValueError(f"{path}: too short for a table footer")
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack("<Q", footer[40:])[0] != MAGIC:
            raise ValueError(f"{path}: bad table magic (not a TensorFlow checkpoint index)")
        pos = 0
        _, pos = _read_varint(footer, pos)          # metaindex handle
        _, pos = _read_varint(footer, pos)
        ioff, pos = _read_varint(footer, pos)
        isz, pos = _read_varint(footer, pos)
        out: Dict[bytes, bytes] = {}
        for _, handle in _block_entries(_read_block(f, ioff, isz, verify)):
            boff, p2 = _read_varint(handle, 0)
            bsz, _ = _read_varint(handle, p2)
            for k, v in _block_entries(_read_block(f, boff, bsz, verify)):
                out[k] = v
    return out


def _build_block(items: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path: str, items: Dict[bytes, bytes], block_bytes: int = 4096) -> None:
    keys = sorted(items)
    with open(path, "wb") as f:
        def emit(block: bytes) -> bytes:
            off = f.tell()
            f.write(block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
            return _varint(off) + _varint(len(block))
        index, cur, cur_size = [], [], 0
        for k in keys:
            cur.append((k, items[k]))
            cur_size += len(k) + len(items[k]) + 8
            if cur_size >= block_bytes:
                index.append((cur[-1][0], emit(_build_block(cur))))
                cur, cur_size = [], 0
        if cur:
            index.append((cur[-1][0], emit(_build_block(cur))))
        meta = emit(_build_block([]))
        idx = emit(_build_block(index, restart_interval=1))
        footer = meta + idx
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))


# ---- tensor bundle -----------------------------------------------------------------------------------------------------------
class BundleReader:
    """``reader.keys()``, ``reader.shape(key)``, ``reader.tensor(key)`` over ``<prefix>.index`` + ``<prefix>.data-*``."""

    def __init__(self, prefix: str, verify_index: bool = True):
        self.prefix = prefix
        table = read_table(prefix + ".index", verify_index)
        if b"" not in table:
            raise ValueError("bundle header entry is missing")
        hdr = parse_message(table[b""])
        self.num_shards = hdr.get(1, [1])[0]
        if hdr.get(2, [0])[0] != 0:
            raise ValueError("big-endian bundles are not supported")
        self.entries: Dict[str, dict] = {}
        for k, v in table.items():
            if k == b"":
                continue
            m = parse_message(v)
            shape = []
            if 2 in m:
                for d in parse_message(m[2][0]).get(2, []):
                    shape.append(parse_message(d).get(1, [0])[0])
            slices = None
            if 7 in m:                                      # a partitioned variable: the pieces live under their own (binary) keys
                slices = []
                for sl in m[7]:
                    ext = []
                    for e in parse_message(sl).get(1, []):
                        em = parse_message(e)
                        ext.append((_sint64(em.get(1, [0])[0]), _sint64(em[2][0]) if 2 in em else -1))      # (start, length); length -1: the whole axis
                    slices.append(tuple(ext))
            try:
                name = k.decode()
            except UnicodeDecodeError:
                name = k.decode("latin-1")                  # slice keys (OrderedCode): kept byte for byte
            self.entries[name] = dict(dtype=m.get(1, [0])[0], shape=tuple(shape), shard=m.get(3, [0])[0],
                                      offset=m.get(4, [0])[0], size=m.get(5, [0])[0], crc=m.get(6, [None])[0], slices=slices)
        self._maps: Dict[int, np.memmap] = {}

    def keys(self) -> List[str]:
        """The variables (slice pieces of partitioned variables are reached through their variable's key)."""
        return sorted(k for k in self.entries if not k.startswith("\x00"))

    def shape(self, key: str) -> Tuple[int, ...]:
        return self.entries[key]["shape"]

    def _shard(self, i: int):
        if i not in self._maps:
            self._maps[i] = np.memmap(f"{self.prefix}.data-{i:05d}-of-{self.num_shards:05d}", dtype=np.uint8, mode="r")
        return self._maps[i]

    def raw(self, key: str) -> bytes:
        e = self.entries[key]
        return bytes(self._shard(e["shard"])[e["offset"]:e["offset"] + e["size"]])

    def tensor(self, key: str, verify: bool = False) -> np.ndarray:
        e = self.entries[key]
        if e["dtype"] not in _NP:
            raise ValueError(f"{key}: dtype {e['dtype']} is not numeric")
        if e["slices"] is not None:
            # tf.Variable partitioned by a partitioner / saved with a SaveSliceInfo: the full-tensor entry lists the slices, every slice is
            # an entry of its own under checkpoint::EncodeTensorNameSlice(name, slice) (tensorflow/core/util/saved_tensor_slice_util.cc)
            out = np.empty(e["shape"], dtype=_NP[e["dtype"]])
            covered = np.zeros(e["shape"], dtype=bool)
            for ext in e["slices"]:
                sk = encode_slice_key(key, ext).decode("latin-1")
                if sk not in self.entries:
                    raise ValueError(f"{key}: slice {ext} is listed but has no entry")
                idx = tuple(slice(None) if ln < 0 else slice(st, st + ln) for st, ln in ext)
                piece = self.tensor(sk, verify)
                if piece.shape != out[idx].shape:
                    raise ValueError(f"{key}: slice {ext} has shape {piece.shape}, expected {out[idx].shape}")
                out[idx] = piece
                covered[idx] = True
            if not covered.all():
                raise ValueError(f"{key}: the saved slices do not cover the variable")
            return out
        buf = self._shard(e["shard"])[e["offset"]:e["offset"] + e["size"]]
        if verify and e["crc"] is not None and mask_crc(crc32c(bytes(buf))) != e["crc"]:
            raise ValueError(f"{key}: tensor checksum mismatch")
        return np.frombuffer(buf, dtype=_NP[e["dtype"]]).reshape(e["shape"]).copy()

    def string_scalar(self, key: str) -> bytes:
        """A scalar DT_STRING tensor: varint64 length, masked crc32c of the length bytes, then the bytes."""
        e = self.entries[key]
        if e["dtype"] != DT_STRING:
            raise ValueError(f"{key} is not a string tensor")
        raw = self.raw(key)
        n, pos = _read_varint(raw, 0)
        return raw[pos + 4:pos + 4 + n]


def write_bundle(prefix: str, tensors: Dict[str, object]) -> None:
    """``tensors``: key -> ndarray (numeric) or bytes (scalar string tensor).  One data shard."""
    items: Dict[bytes, bytes] = {b"": _f_varint(1, 1) + _f_bytes(3, _f_varint(1, 1))}      # num_shards = 1, version.producer = 1
    off = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for key in sorted(tensors):
            val = tensors[key]
            if isinstance(val, (bytes, bytearray)):
                ln = _varint(len(val))
                payload = ln + struct.pack("<I", mask_crc(crc32c(ln))) + bytes(val)
                dtype, shape = DT_STRING, ()
                crc = mask_crc(crc32c(bytes(val), crc32c(ln)))
            else:
                arr = np.asarray(val)                                  # (ascontiguousarray would turn a scalar into shape (1,))
                payload, dtype, shape = arr.tobytes(), _DT[arr.dtype], arr.shape
                crc = mask_crc(crc32c(payload)) if arr.nbytes <= (1 << 20) else 0      # pure-Python CRC: skipped for big tensors
            f.write(payload)
            shp = b"".join(_f_bytes(2, _f_varint(1, d)) for d in shape)
            entry = _f_varint(1, dtype) + _f_bytes(2, shp) + _f_varint(4, off) + _f_varint(5, len(payload)) + _field(6, 5, struct.pack("<I", crc))
            items[key.encode()] = entry
            off += len(payload)
    write_table(prefix + ".index", items)


# ---- trackable object graph --------------------------------------------------------------------------------------------------
class ObjectGraph:
    """nodes[i] = {"children": {local_name: node_id}, "vars": {attr_name: (full_name, checkpoint_key)}}"""

    def __init__(self, blob: bytes):
        self.nodes = []
        for nb in parse_message(blob).get(1, []):
            m = parse_message(nb)
            children = {}
            for c in m.get(1, []):
                cm = parse_message(c)
                children[cm.get(2, [b""])[0].decode()] = cm.get(1, [0])[0]
            attrs = {}
            for a in m.get(2, []):
                am = parse_message(a)
                attrs[am.get(1, [b""])[0].decode()] = (am.get(2, [b""])[0].decode(), am.get(3, [b""])[0].decode())
            slots = []                                   # SlotVariableReference {original_variable_node_id = 1, slot_name = 2, slot_variable_node_id = 3}
            for sr in m.get(3, []):
                sm = parse_message(sr)
                slots.append((sm.get(1, [0])[0], sm.get(2, [b""])[0].decode(), sm.get(3, [0])[0]))
            self.nodes.append({"children": children, "vars": attrs, "slots": slots})

    def child(self, node: int, *path: str) -> int:
        for p in path:
            node = self.nodes[node]["children"][p]
        return node

    def variable(self, node: int) -> Optional[Tuple[str, str]]:
        return self.nodes[node]["vars"].get("VARIABLE_VALUE")


def serialize_object_graph(nodes: List[dict]) -> bytes:
    out = b""
    for n in nodes:
        body = b""
        for name, nid in n.get("children", {}).items():
            body += _f_bytes(1, _f_varint(1, nid) + _f_bytes(2, name.encode()))
        for attr, (full, key) in n.get("vars", {}).items():
            body += _f_bytes(2, _f_bytes(1, attr.encode()) + _f_bytes(2, full.encode()) + _f_bytes(3, key.encode()))
        for orig, slot, node in n.get("slots", []):
            body += _f_bytes(3, _f_varint(1, orig) + _f_bytes(2, slot.encode()) + _f_varint(3, node))
        out += _f_bytes(1, body)
    return out


# ---- Keras naming of the reference's U-Net -----------------------------------------------------------------------------------
_ATTN_ATTRS = {  # attribute path inside CrossAttentionBlock (conditional_dm3d.py:120-137) -> this package's sub-name
    ("norm",): "norm", ("norm1",): "ln1", ("norm2",): "ln2", ("norm3",): "ln3", ("proj_in",): "proj_in", ("proj_out",): "proj_out",
    ("query",): "query", ("key",): "key", ("value",): "value", ("proj", 0): "mlp.0", ("proj", 1): "mlp.1",
}
_SELF_ATTRS = {("norm",): "norm", ("query",): "query", ("key",): "key", ("value",): "value", ("proj",): "proj"}   # dm3d.py:18-38
_VAR = {"kernel": "kernel", "bias": "bias", "gamma": "gamma", "beta": "beta", "moving_mean": "mean", "moving_variance": "var",
        "embeddings": "table"}


def keras_layer_plan(cfg) -> List[Tuple[str, str]]:
    """Top-level weighted Keras layers of build_model in CREATION order as (keras class prefix, this package's name):
    class prefix is what Keras derives its auto names from (conv3d, dense, batch_normalization, embedding,
    cross_attention_block / attention_block)."""
    from .weights import walk
    blocks, _ = walk(cfg)
    plan = [("conv3d", "conv_in"), ("dense", "time_mlp.0"), ("dense", "time_mlp.1")]
    if cfg.conditional:
        plan.append(("embedding", "ctx_embed"))
    def res(name, cin, width):
        if cin != width:
            plan.append(("conv3d", f"{name}.skip"))
        plan.extend([("dense", f"{name}.temb"), ("batch_normalization", f"{name}.norm1"), ("conv3d", f"{name}.conv1"),
                     ("batch_normalization", f"{name}.norm2"), ("conv3d", f"{name}.conv2")])

    for blk in blocks:
        if blk.kind == "res":
            res(blk.name, blk.cin + blk.cskip, blk.cout)
        elif blk.kind == "attn":
            if cfg.conditional:
                plan.append(("dense", f"{blk.name}.ctx_mlp"))                  # ContextMLP is built before the block (:371-372)
                plan.append(("cross_attention_block", blk.name))
            else:
                plan.append(("attention_block", blk.name))
        elif blk.kind in ("down", "up"):
            plan.append(("conv3d", blk.name))
    plan.extend([("batch_normalization", "out.norm"), ("conv3d", "out.conv")])
    return plan


def _auto_index(full_name: str) -> int:
    """Creation number Keras gave the layer that owns a variable: 'conv3d_12/kernel' -> 12, 'dense/bias' -> 0,
    'cross_attention_block_1/dense_40/kernel' -> 40 (the component right before the variable name)."""
    parts = full_name.split("/")
    owner = parts[-2] if len(parts) >= 2 else parts[0]
    m = re.match(r"^.*?(?:_(\d+))?$", owner)
    return int(m.group(1) or 0)


def _classify(vars_, shapes) -> Tuple[str, int]:
    """(keras class prefix, creation number) of one top-level layer from its variables [(path, attr, full_name, key)]:
    structure first (an attention block is recognised by its query sublayer, whatever its variables are called), then the
    variable kinds.  Blocks sort by the creation number of their query Dense, which is drawn from the same per-class counter
    in constructor order."""
    paths = {p for p, _, _, _ in vars_}
    if ("query",) in paths:
        q = next(full for p, a, full, _ in vars_ if p == ("query",) and a == "kernel")
        return ("cross_attention_block" if ("norm1",) in paths else "attention_block"), _auto_index(q)
    attrs = {a for _, a, _, _ in vars_}
    first = vars_[0]
    if "embeddings" in attrs:
        return "embedding", _auto_index(first[2])
    if "moving_mean" in attrs:
        return "batch_normalization", _auto_index(first[2])
    if "kernel" in attrs:
        k = next(key for _, a, _, key in vars_ if a == "kernel")
        return ("conv3d" if len(shapes(k)) == 5 else "dense"), _auto_index(first[2])
    raise ValueError(f"cannot classify checkpoint layer with variables {sorted(attrs)}")


def _collect(graph: ObjectGraph, node: int, path=()):
    """(attribute path, variable attr name, checkpoint_key) for every variable under ``node`` (layers / layer_with_weights-i
    children of a Sequential become integer path components)."""
    out = []
    for name, nid in graph.nodes[node]["children"].items():
        var = graph.variable(nid)
        if var is not None:
            out.append((path, name, var[0], var[1]))
            continue
        m = re.match(r"^layer_with_weights-(\d+)$", name)
        if m:
            out.extend(_collect(graph, nid, path + (int(m.group(1)),)))
        elif not re.match(r"^(layer-\d+|layers|_.*|keras_api|optimizer|loss_tracker|metrics|variables|trainable_variables|"
                          r"non_trainable_variables|regularization_losses|layer_metrics|layer_regularization_losses)$", name):
            out.extend(_collect(graph, nid, path + (name,)))
    return out


def load_unet_state(prefix: str, cfg, root: Tuple[str, ...] = ("network",), verify: bool = False, with_optimizer: bool = False) -> Dict[str, np.ndarray]:
    """Reads the U-Net weights of a reference checkpoint (``DiffusionModel.save_weights(prefix)``; ``root=()`` for a bare
    ``network.save_weights``) into this package's state dict.  Raises with the offending names on any count or shape mismatch.
    ``with_optimizer``: also the Adam state of a compiled model's checkpoint (``optimizer`` child of the root with ``iter`` and the m / v
    slot variables, referenced through the object graph's slot_variables — the layout OptimizerV2 writes, TF <= 2.10 and
    tf.keras.optimizers.legacy) as ``optimizer/iter``, ``optimizer/m/<name>``, ``optimizer/v/<name>``: what ``model.load_weights(<epoch>.ckpt)``
    resumes with in the reference (main_conditional_dm.py:174-183).  A checkpoint without them yields the weights alone."""
    from .weights import param_spec
    rd = BundleReader(prefix)
    if OBJECT_GRAPH_KEY not in rd.entries:
        raise ValueError("not an object-based TF2 checkpoint (no _CHECKPOINTABLE_OBJECT_GRAPH); name-based V1 checkpoints "
                         "carry Keras layer names directly and are not produced by the reference")
    graph = ObjectGraph(rd.string_scalar(OBJECT_GRAPH_KEY))
    try:
        net = graph.child(0, *root)
    except KeyError as e:
        raise ValueError(f"object {'/'.join(root)} not found under the checkpoint root (children: "
                         f"{sorted(graph.nodes[0]['children'])})") from e
    layers = {}                                            # top-level layer node -> its variables
    for name, nid in graph.nodes[net]["children"].items():
        if re.match(r"^layer_with_weights-\d+$", name):
            layers[nid] = _collect(graph, nid)
    by_class: Dict[str, List[Tuple[int, int]]] = {}
    for nid, vars_ in layers.items():
        if not vars_:
            continue
        cls, idx = _classify(vars_, rd.shape)
        by_class.setdefault(cls, []).append((idx, nid))
    plan = keras_layer_plan(cfg)
    want: Dict[str, List[str]] = {}
    for cls, name in plan:
        want.setdefault(cls, []).append(name)
    spec = param_spec(cfg)
    state: Dict[str, np.ndarray] = {}
    key_of_target: Dict[str, str] = {}
    for cls, names in want.items():
        have = sorted(by_class.get(cls, []))
        if len(have) != len(names):
            raise ValueError(f"checkpoint has {len(have)} '{cls}' layers under {'/'.join(root) or '<root>'}, the configuration "
                             f"builds {len(names)}: widths / has_attention / num_res_blocks differ from the saved model")
        for (_, nid), name in zip(have, names):
            for path, attr, _full, key in layers[nid]:
                if cls in ("cross_attention_block", "attention_block"):
                    table = _ATTN_ATTRS if cls == "cross_attention_block" else _SELF_ATTRS
                    if path not in table:
                        raise ValueError(f"unexpected sublayer {path} in {cls} {name}")
                    target = f"{name}.{table[path]}.{_VAR[attr]}"
                else:
                    target = f"{name}.{_VAR[attr]}"
                arr = rd.tensor(key, verify)
                key_of_target[target] = key
                if target not in spec:
                    raise ValueError(f"checkpoint variable {key} maps to unknown parameter {target}")
                if tuple(arr.shape) != tuple(spec[target]):
                    raise ValueError(f"{target}: checkpoint shape {arr.shape} != expected {spec[target]} ({key})")
                state[target] = arr.astype(np.float32, copy=False)
    missing = [n for n in spec if n not in state]
    if missing:
        raise ValueError(f"checkpoint lacks {len(missing)} parameters, e.g. {missing[:4]}")
    if with_optimizer and "optimizer" in graph.nodes[0]["children"]:
        opt = graph.nodes[0]["children"]["optimizer"]
        node_of_key = {v["vars"]["VARIABLE_VALUE"][1]: i for i, v in enumerate(graph.nodes) if "VARIABLE_VALUE" in v["vars"]}
        target_of_node = {node_of_key[key]: target for target, key in key_of_target.items() if key in node_of_key}
        it = graph.nodes[opt]["children"].get("iter")
        if it is not None and graph.variable(it) is not None:
            state["optimizer/iter"] = np.asarray(rd.tensor(graph.variable(it)[1], verify), np.int64).reshape(())
        for orig, slot, node in graph.nodes[opt]["slots"]:
            if orig in target_of_node and slot in ("m", "v") and graph.variable(node) is not None:
                state[f"optimizer/{slot}/{target_of_node[orig]}"] = rd.tensor(graph.variable(node)[1], verify).astype(np.float32, copy=False)
    return state


def save_unet_checkpoint(prefix: str, state: Dict[str, np.ndarray], cfg, root: Tuple[str, ...] = ("network",),
                         first_index: Optional[Dict[str, int]] = None, shuffle_seed: Optional[int] = None,
                         optimizer: Optional[Dict[str, np.ndarray]] = None) -> None:
    """Writes ``state`` as an object-based checkpoint laid out the way Keras would for the reference's model: auto layer names
    numbered per class in creation order (``first_index`` = numbers already used by models built earlier in the process, e.g.
    the VQ-VAE), ``layer_with_weights-N`` indices in an order unrelated to creation order (``shuffle_seed``), custom-layer
    sublayers under their attribute names.  Lets weights trained here flow back into the reference, and is the fixture
    generator of the tests."""
    plan = keras_layer_plan(cfg)
    counters = dict(first_index or {})                      # per-class creation counters; sublayers draw from the same ones

    def auto(cls):
        i = counters.get(cls, 0)
        counters[cls] = i + 1
        return cls if i == 0 else f"{cls}_{i}"

    inv_var = {v: k for k, v in _VAR.items()}
    order = list(range(len(plan)))
    if shuffle_seed is not None:
        np.random.default_rng(shuffle_seed).shuffle(order)
    lw_index = {p: i for i, p in enumerate(order)}          # plan position -> layer_with_weights number
    nodes: List[dict] = [{"children": {}, "vars": {}}]
    tensors: Dict[str, object] = {}

    def new_node():
        nodes.append({"children": {}, "vars": {}})
        return len(nodes) - 1

    parent = 0
    prefix_key = ""
    for r in root:
        n = new_node()
        nodes[parent]["children"][r] = n
        parent, prefix_key = n, prefix_key + r + "/"

    var_nodes: Dict[int, Tuple[int, str, str]] = {}          # id(array of ``state``) -> (node, key path, keras full name)

    def add_var(owner, owner_key, attr, full, arr):
        v = new_node()
        nodes[owner]["children"][attr] = v
        key = f"{owner_key}/{attr}{VALUE_SUFFIX}"
        nodes[v]["vars"]["VARIABLE_VALUE"] = (full, key)
        tensors[key] = np.asarray(arr, np.float32)
        var_nodes[id(arr)] = (v, f"{owner_key}/{attr}", full)

    def params_of(name):
        return [(k[len(name) + 1:], v) for k, v in state.items() if k.startswith(name + ".") and "." not in k[len(name) + 1:]]

    for pos, (cls, name) in enumerate(plan):
        layer = new_node()
        lkey = f"{prefix_key}layer_with_weights-{lw_index[pos]}"
        nodes[parent]["children"][f"layer_with_weights-{lw_index[pos]}"] = layer
        lname = auto(cls)
        if cls in ("cross_attention_block", "attention_block"):
            table = _ATTN_ATTRS if cls == "cross_attention_block" else _SELF_ATTRS
            sub_cls = {"norm": "batch_normalization", "ln1": "layer_normalization", "ln2": "layer_normalization",
                       "ln3": "layer_normalization", "proj_in": "conv3d", "proj_out": "conv3d"}
            seq_node = None
            for path, sub in table.items():                 # dict order = the reference's constructor order
                sname = auto(sub_cls.get(sub, "dense"))
                if len(path) == 2 and isinstance(path[1], int):
                    if seq_node is None:
                        seq_node = new_node()
                        nodes[layer]["children"][path[0]] = seq_node
                    owner = new_node()
                    nodes[seq_node]["children"][f"layer_with_weights-{path[1]}"] = owner
                    okey = f"{lkey}/{path[0]}/layer_with_weights-{path[1]}"
                else:
                    owner = new_node()
                    nodes[layer]["children"][path[0]] = owner
                    okey = f"{lkey}/{path[0]}"
                for short, arr in params_of(f"{name}.{sub}"):
                    add_var(owner, okey, inv_var[short], f"{lname}/{sname}/{inv_var[short]}", arr)
        else:
            for short, arr in params_of(name):
                add_var(layer, lkey, inv_var[short], f"{lname}/{inv_var[short]}", arr)
    # ``optimizer``: {"optimizer/iter", "optimizer/m/<name>", "optimizer/v/<name>"} (Trainer.optimizer_state()): the Adam state as
    # OptimizerV2 checkpoints it — an ``optimizer`` child of the root with ``iter``, and one slot variable per (trainable variable, slot)
    # under <variable key>/.OPTIMIZER_SLOT/optimizer/<slot>, referenced from the optimizer object's slot_variables
    if optimizer:
        opt = new_node()
        nodes[0]["children"]["optimizer"] = opt
        itn = new_node()
        nodes[opt]["children"]["iter"] = itn
        nodes[itn]["vars"]["VARIABLE_VALUE"] = ("Adam/iter", f"optimizer/iter{VALUE_SUFFIX}")
        tensors[f"optimizer/iter{VALUE_SUFFIX}"] = np.asarray(optimizer["optimizer/iter"], np.int64).reshape(())
        nodes[opt]["slots"] = []
        for name, arr in state.items():
            if id(arr) not in var_nodes:
                continue
            v, kpath, full = var_nodes[id(arr)]
            for slot in ("m", "v"):
                if f"optimizer/{slot}/{name}" not in optimizer:
                    continue
                sn = new_node()
                key = f"{kpath}/.OPTIMIZER_SLOT/optimizer/{slot}{VALUE_SUFFIX}"
                nodes[sn]["vars"]["VARIABLE_VALUE"] = (f"Adam/{full}/{slot}", key)
                tensors[key] = np.asarray(optimizer[f"optimizer/{slot}/{name}"], np.float32)
                nodes[opt]["slots"].append((v, slot, sn))
    tensors[OBJECT_GRAPH_KEY] = serialize_object_graph(nodes)
    write_bundle(prefix, tensors)


# ---- the autoencoder (networks/vqvae3d_monai.py): nested keras.Model subclasses, mapped purely by creation order per class ------
_VQ_ATTR = {"kernel": "kernel", "bias": "bias", "gamma": "gamma", "beta": "beta", "moving_mean": "mean", "moving_variance": "var",
            "alpha": "alpha"}


def _vq_layer_plan(spec: Dict[str, tuple]) -> Dict[str, List[str]]:
    """keras class prefix -> this package's layer names in creation order (= ``vqvae_param_spec`` order: Encoder, then Decoder;
    the reference constructs encoder, decoder, quantizer in that order, vqvae3d_monai.py:418-442)."""
    plan: Dict[str, List[str]] = {"conv3d": [], "conv3d_transpose": [], "batch_normalization": [], "p_re_lu": []}
    for name in spec:
        if name.endswith(".kernel"):
            base = name[:-7]
            plan["conv3d_transpose" if base.startswith("dec.up") else "conv3d"].append(base)
        elif name.endswith(".gamma"):
            plan["batch_normalization"].append(name[:-6])
        elif name.endswith(".alpha"):
            plan["p_re_lu"].append(name[:-6])
    return plan


def _all_variables(graph: ObjectGraph, node: int):
    """Every (full_name, checkpoint_key) reachable from ``node`` (aliases such as layer-3 / layer_with_weights-1 visit a node once)."""
    seen, out, stack = set(), {}, [node]
    while stack:
        n = stack.pop()
        if n in seen:
            continue
        seen.add(n)
        var = graph.variable(n)
        if var is not None:
            out[var[1]] = var[0]
        stack.extend(graph.nodes[n]["children"].values())
    return [(full, key) for key, full in out.items()]


def load_vqvae_state(prefix: str, spec: Dict[str, tuple], root: Tuple[str, ...] = (), verify: bool = False,
                     parts: Optional[Tuple[str, ...]] = None) -> Dict[str, np.ndarray]:
    """Autoencoder weights of a reference checkpoint (``vqvae.save_weights``: root (); inside a DiffusionModel checkpoint:
    root ("vqvae_trainer",)) -> the state dict of ``networks.vqvae3d_monai.VQVAE``.  ``parts``: only these children of the saved object
    are read — ("encoder", "decoder", "quantizer") for ``networks/vqgan.py``, whose checkpoints also carry two discriminators (Conv3D /
    Conv2D / Dense layers created AFTER the autoencoder: their creation numbers follow the autoencoder's), the LPIPS network, metric
    trackers and the optimizers' slots (vqgan.py:644-696)."""
    rd = BundleReader(prefix)
    if OBJECT_GRAPH_KEY not in rd.entries:
        raise ValueError("not an object-based TF2 checkpoint (no _CHECKPOINTABLE_OBJECT_GRAPH)")
    graph = ObjectGraph(rd.string_scalar(OBJECT_GRAPH_KEY))
    try:
        top = graph.child(0, *root)
    except KeyError as e:
        raise ValueError(f"object {'/'.join(root)} not found under the checkpoint root") from e
    layers: Dict[Tuple[str, int], Dict[str, str]] = {}
    state: Dict[str, np.ndarray] = {}
    if parts is None:
        found = _all_variables(graph, top)
    else:
        missing_parts = [q for q in parts if q not in graph.nodes[top]["children"]]
        if missing_parts:
            raise ValueError(f"checkpoint object has no {missing_parts} (children: {sorted(graph.nodes[top]['children'])})")
        found = [v for q in parts for v in _all_variables(graph, graph.nodes[top]["children"][q])]
    for full, key in found:
        parts = full.split("/")
        attr = parts[-1].split(":")[0]
        if attr.startswith("embeddings_vqvae"):
            state["vq.embeddings"] = rd.tensor(key, verify).astype(np.float32, copy=False)
            continue
        if attr not in _VQ_ATTR or len(parts) < 2:
            continue                                        # optimizer slots, metric totals, codebooks_used ...
        m = re.match(r"^(.*?)(?:_(\d+))?$", parts[-2])
        layers.setdefault((m.group(1), int(m.group(2) or 0)), {})[attr] = key
    plan = _vq_layer_plan(spec)
    for cls, names in plan.items():
        have = sorted(k for k in layers if k[0] == cls)
        if len(have) != len(names):
            raise ValueError(f"checkpoint has {len(have)} '{cls}' layers, the configuration builds {len(names)}")
        for k, name in zip(have, names):
            for attr, key in layers[k].items():
                target = f"{name}.{_VQ_ATTR[attr]}"
                arr = rd.tensor(key, verify)
                if target not in spec or tuple(arr.shape) != tuple(spec[target]):
                    raise ValueError(f"{target}: checkpoint shape {arr.shape} != expected {spec.get(target)} ({key})")
                state[target] = arr.astype(np.float32, copy=False)
    missing = [n for n in spec if n not in state]
    if missing:
        raise ValueError(f"checkpoint lacks {len(missing)} parameters, e.g. {missing[:4]}")
    return state


def save_vqvae_checkpoint(prefix: str, state: Dict[str, np.ndarray], spec: Dict[str, tuple], root: Tuple[str, ...] = (),
                          first_index: Optional[Dict[str, int]] = None, extra: Optional[Dict[str, np.ndarray]] = None) -> None:
    """Writes the autoencoder ``state`` with Keras-style names (auto layer names per class in creation order, nested under
    encoder / decoder / quantizer objects).  Export path and test-fixture generator."""
    counters = dict(first_index or {})
    nodes: List[dict] = [{"children": {}, "vars": {}}]
    tensors: Dict[str, object] = {}

    def new_node(parent, name):
        nodes.append({"children": {}, "vars": {}})
        nodes[parent]["children"][name] = len(nodes) - 1
        return len(nodes) - 1

    top, key_prefix = 0, ""
    for r in root:
        top, key_prefix = new_node(top, r), key_prefix + r + "/"
    parts = {"enc": new_node(top, "encoder"), "dec": new_node(top, "decoder"), "vq": new_node(top, "quantizer")}
    part_key = {"enc": "encoder", "dec": "decoder", "vq": "quantizer"}
    inv = {v: k for k, v in _VQ_ATTR.items()}
    count = {"enc": 0, "dec": 0}
    plan = _vq_layer_plan(spec)
    cls_of = {name: cls for cls, names in plan.items() for name in names}
    done = set()
    for pname in spec:                                       # spec order = creation order
        if pname == "vq.embeddings":
            v = new_node(parts["vq"], "embeddings")
            key = f"{key_prefix}quantizer/embeddings{VALUE_SUFFIX}"
            nodes[v]["vars"]["VARIABLE_VALUE"] = ("embeddings_vqvae", key)
            tensors[key] = np.asarray(state[pname], np.float32)
            continue
        base = pname.rsplit(".", 1)[0]
        if base in done:
            continue
        done.add(base)
        cls = cls_of[base]
        i = counters.get(cls, 0)
        counters[cls] = i + 1
        lname = cls if i == 0 else f"{cls}_{i}"
        part = base.split(".")[0]
        layer = new_node(parts[part], f"layer_with_weights-{count[part]}")
        lkey = f"{key_prefix}{part_key[part]}/layer_with_weights-{count[part]}"
        count[part] += 1
        for k in spec:
            if k.rsplit(".", 1)[0] == base:
                attr = inv[k.rsplit(".", 1)[1]]
                v = new_node(layer, attr)
                key = f"{lkey}/{attr}{VALUE_SUFFIX}"
                nodes[v]["vars"]["VARIABLE_VALUE"] = (f"vqvae/{part_key[part]}/{lname}/{attr}", key)
                tensors[key] = np.asarray(state[k], np.float32)
    # ``extra``: variables of sibling objects a real checkpoint holds beside the autoencoder (the VQGAN's discriminators): name
    # "discriminator/conv3d_40/kernel" -> child "discriminator", layer_with_weights-k, Keras full name as given
    for j, (full, arr) in enumerate((extra or {}).items()):
        obj, lname, attr = full.split("/")
        onode = nodes[top]["children"].get(obj) or new_node(top, obj)
        layer = new_node(onode, f"layer_with_weights-{j}")
        v = new_node(layer, attr)
        key = f"{key_prefix}{obj}/layer_with_weights-{j}/{attr}{VALUE_SUFFIX}"
        nodes[v]["vars"]["VARIABLE_VALUE"] = (f"{lname}/{attr}", key)
        tensors[key] = np.asarray(arr, np.float32)
    tensors[OBJECT_GRAPH_KEY] = serialize_object_graph(nodes)
    write_bundle(prefix, tensors)
