"""Hand-assembled TensorFlow tensor-bundle fixtures (tests/golden/tf_*): bytes laid out from the published formats — the LevelDB table
format (doc/table_format.md: blocks of prefix-compressed entries with a restart array, 5-byte block trailer, index block of shortest
separators, 48-byte footer), tensorflow/core/util/tensor_bundle (BundleHeaderProto / BundleEntryProto, several data shards, string
tensors), saved_tensor_slice_util.cc + lib/strings/ordered_code.cc (partitioned variables) — and NOT through the package's own writer
(3d-condtional-stable-diffusion_amd/tf_checkpoint.py): nothing below imports it, the CRC-32C is a bit-serial implementation of its own, and the files exercise what
TensorFlow emits and that writer does not: three `.data-0000x-of-00003` shards with non-zero offsets, a variable saved in slices
(along axis 0 and along axis 1), an index of several data blocks with restart interval 16 and long shared key prefixes, shortest-separator
index keys.  No TensorFlow is installed here and the reference ships no checkpoint: these are format fixtures, not
outputs of the reference.  The tensors' values are a function of their key (expected() below), so the test recomputes them.

    python tests/golden/make_tf_fixtures.py          (rewrites tests/golden/tf_multi.* and tests/golden/tf_sliced.*)
"""
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAGIC = 0xDB4775248B80FB57


def crc32c_bitwise(data: bytes) -> int:
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), one bit at a time."""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc ^ 0xFFFFFFFF


def masked(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def pb_varint(field, v): return varint(field << 3) + varint(v)
def pb_bytes(field, b): return varint((field << 3) | 2) + varint(len(b)) + b
def pb_fixed32(field, v): return varint((field << 3) | 5) + struct.pack("<I", v)


def expected(key: str, shape, dtype=np.float32) -> np.ndarray:
    """The value every fixture tensor holds: a ramp offset by a checksum of its key."""
    n = int(np.prod(shape)) if shape else 1
    base = zlib.crc32(key.encode()) % 97
    return (np.arange(n, dtype=np.float64) * 0.25 + base).astype(dtype).reshape(shape)


# ---- table ------------------------------------------------------------------------------------------------------------------------
def build_block(entries, restart_interval):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts or [0]))
    return bytes(out)


def shortest_separator(a: bytes, b: bytes) -> bytes:
    """leveldb BytewiseComparator::FindShortestSeparator: a <= result < b, as short as possible"""
    n = min(len(a), len(b))
    i = 0
    while i < n and a[i] == b[i]:
        i += 1
    if i < n and a[i] < 0xFF and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def short_successor(a: bytes) -> bytes:
    for i, c in enumerate(a):
        if c != 0xFF:
            return a[:i] + bytes([c + 1])
    return a


def write_table(path, items, block_bytes, restart_interval=16):
    keys = sorted(items)
    blocks, cur, size = [], [], 0
    for k in keys:
        cur.append((k, items[k]))
        size += len(k) + len(items[k])
        if size >= block_bytes:
            blocks.append(cur)
            cur, size = [], 0
    if cur:
        blocks.append(cur)
    with open(path, "wb") as f:
        def emit(block):
            off = f.tell()
            f.write(block + b"\x00" + struct.pack("<I", masked(crc32c_bitwise(block + b"\x00"))))      # type 0 = no compression
            return varint(off) + varint(len(block))
        handles = [emit(build_block(b, restart_interval)) for b in blocks]
        index = []
        for i, b in enumerate(blocks):
            last = b[-1][0]
            sep = shortest_separator(last, blocks[i + 1][0][0]) if i + 1 < len(blocks) else short_successor(last)
            index.append((sep, handles[i]))
        meta = emit(build_block([], restart_interval))
        idx = emit(build_block(index, 1))
        footer = meta + idx
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    return len(blocks)


# ---- bundle -----------------------------------------------------------------------------------------------------------------------
def shape_proto(shape): return b"".join(pb_bytes(2, pb_varint(1, d)) for d in shape)


def entry(dtype, shape, shard, offset, size, crc):
    e = pb_varint(1, dtype) + pb_bytes(2, shape_proto(shape))
    if shard:
        e += pb_varint(3, shard)
    if offset:
        e += pb_varint(4, offset)
    return e + pb_varint(5, size) + pb_fixed32(6, crc)


def header(num_shards): return pb_varint(1, num_shards) + pb_bytes(3, pb_varint(1, 1))      # endianness little (default), version.producer 1


def oc_num(v):
    body = v.to_bytes((v.bit_length() + 7) // 8, "big") if v else b""
    return bytes([len(body)]) + body


def oc_str(b): return b"".join(b"\x00\xff" if c == 0 else (b"\xff\x00" if c == 255 else bytes([c])) for c in b) + b"\x00\x01"


def oc_signed(v):
    x = ~v if v < 0 else v
    if x < 64:
        return bytes([(0x80 ^ v) & 0xFF])
    n = 2
    while 7 * n - 1 < x.bit_length():
        n += 1
    buf = bytearray((v & ((1 << 80) - 1)).to_bytes(10, "big"))
    hdr = [(0, 0), (0x80, 0), (0xC0, 0), (0xE0, 0), (0xF0, 0), (0xF8, 0), (0xFC, 0), (0xFE, 0), (0xFF, 0), (0xFF, 0x80), (0xFF, 0xC0)][n]
    buf[10 - n] ^= hdr[0]
    buf[11 - n] ^= hdr[1]
    return bytes(buf[10 - n:])


def slice_key(name, extents):
    out = oc_num(0) + oc_str(name.encode()) + oc_num(len(extents))
    for st, ln in extents:
        out += oc_signed(-1 if ln < 0 else st) + oc_signed(ln)
    return out


def slice_proto(extents):
    out = b""
    for st, ln in extents:
        e = b""
        if ln >= 0:
            if st:
                e += pb_varint(1, st)
            e += pb_varint(2, ln)
        out += pb_bytes(1, e)
    return out


DT_FLOAT, DT_INT32, DT_STRING, DT_INT64 = 1, 3, 7, 9


def multi_shard_fixture():
    """62 variables with Keras-style keys over three data shards, plus a scalar string tensor and an int64 scalar."""
    names = []
    for i in range(20):
        for attr in ("kernel", "bias"):
            names.append(f"network/layer_with_weights-{i}/{attr}/.ATTRIBUTES/VARIABLE_VALUE")
    for i in range(10):
        for attr in ("gamma", "beta"):
            names.append(f"network/layer_with_weights-{20 + i}/norm/{attr}/.ATTRIBUTES/VARIABLE_VALUE")
    names += ["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE", "save_counter/.ATTRIBUTES/VARIABLE_VALUE"]
    shapes = {}
    for j, n in enumerate(names):
        if "/kernel/" in n:
            shapes[n] = (3, 1 + j % 4, 2 + j % 3)
        elif n.startswith(("optimizer", "save_counter")):
            shapes[n] = ()
        else:
            shapes[n] = (2 + j % 5,)
    shards = [bytearray(b"\x00" * 24), bytearray(), bytearray(b"pad!")]          # non-zero first offsets in shards 0 and 2
    items = {b"": header(3)}
    for j, n in enumerate(sorted(names)):
        sh = j % 3
        if shapes[n] == ():
            arr, dt = np.asarray(expected(n, (), np.int64)), DT_INT64
        else:
            arr, dt = expected(n, shapes[n]), DT_FLOAT
        payload = arr.tobytes()
        items[n.encode()] = entry(dt, shapes[n], sh, len(shards[sh]), len(payload), masked(crc32c_bitwise(payload)))
        shards[sh] += payload
    graph = b"object graph bytes \x00\x01\xff stand-in"                            # a DT_STRING scalar: varint length, masked crc of it, bytes
    ln = varint(len(graph))
    payload = ln + struct.pack("<I", masked(crc32c_bitwise(ln))) + graph
    items[b"_CHECKPOINTABLE_OBJECT_GRAPH"] = entry(DT_STRING, (), 1, len(shards[1]), len(payload), masked(crc32c_bitwise(ln + graph)))
    shards[1] += payload
    prefix = os.path.join(HERE, "tf_multi")
    for i, b in enumerate(shards):
        open(f"{prefix}.data-{i:05d}-of-00003", "wb").write(bytes(b))
    nblocks = write_table(prefix + ".index", items, block_bytes=1500)
    return nblocks, len(names)


def sliced_fixture():
    """embedding/table [10, 6] saved in three row slices, dense/kernel [4, 9] saved in two column slices, one ordinary variable."""
    shard = bytearray()
    items = {b"": header(1)}

    def put(key: bytes, dtype, shape, arr):
        payload = arr.tobytes()
        items[key] = entry(dtype, shape, 0, len(shard), len(payload), masked(crc32c_bitwise(payload)))
        shard.extend(payload)

    full = expected("embedding/table", (10, 6))
    parts = [((0, 4), (0, -1)), ((4, 4), (0, -1)), ((8, 2), (0, -1))]
    items[b"embedding/table"] = pb_varint(1, DT_FLOAT) + pb_bytes(2, shape_proto((10, 6))) + b"".join(pb_bytes(7, slice_proto(e)) for e in parts)
    for ext in parts:
        (st, ln), _ = ext
        put(slice_key("embedding/table", ext), DT_FLOAT, (ln, 6), np.ascontiguousarray(full[st:st + ln]))
    full2 = expected("dense/kernel", (4, 9))
    parts2 = [((0, -1), (0, 5)), ((0, -1), (5, 4))]
    items[b"dense/kernel"] = pb_varint(1, DT_FLOAT) + pb_bytes(2, shape_proto((4, 9))) + b"".join(pb_bytes(7, slice_proto(e)) for e in parts2)
    for ext in parts2:
        _, (st, ln) = ext
        put(slice_key("dense/kernel", ext), DT_FLOAT, (4, ln), np.ascontiguousarray(full2[:, st:st + ln]))
    put(b"dense/bias", DT_FLOAT, (9,), expected("dense/bias", (9,)))
    prefix = os.path.join(HERE, "tf_sliced")
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(shard))
    write_table(prefix + ".index", items, block_bytes=1 << 20)


if __name__ == "__main__":
    nb, nv = multi_shard_fixture()
    sliced_fixture()
    print(f"tf_multi: {nv} variables, {nb} index data blocks, 3 shards; tf_sliced: 2 partitioned variables + 1 plain")
