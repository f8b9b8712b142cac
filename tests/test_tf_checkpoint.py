"""TF2 object-based checkpoint reader / writer.  No TensorFlow and no reference checkpoint exist here, so
these tests pin the module against itself and against the published format constants: table magic, masked CRC32C, the
BundleEntryProto field numbers, Keras' per-class creation numbering."""
import numpy as np
import pytest


def test_crc32c_known_answers():
    from dm3d_amd import tf_checkpoint as tc
    assert tc.crc32c(b"") == 0
    assert tc.crc32c(b"123456789") == 0xE3069283            # the CRC-32C check value
    assert tc.crc32c(b"\x00" * 32) == 0x8A9136AA            # RFC 3720 B.4 test vector
    assert tc.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tc.mask_crc(0) == 0xA282EAD8


def test_table_and_bundle_roundtrip(tmp_path):
    from dm3d_amd import tf_checkpoint as tc
    rng = np.random.default_rng(0)
    tensors = {f"layer_with_weights-{i}/kernel/.ATTRIBUTES/VARIABLE_VALUE": rng.normal(size=(3, 3, 3, 4, i + 1)).astype(np.float32)
               for i in range(40)}                           # enough keys for several index blocks and shared prefixes
    tensors["step"] = np.array(7, np.int64)
    tensors["a/string"] = b"hello \x00 world"
    pre = str(tmp_path / "ck")
    tc.write_bundle(pre, tensors)
    raw = open(pre + ".index", "rb").read()
    assert raw[-8:] == (0xDB4775248B80FB57).to_bytes(8, "little")
    rd = tc.BundleReader(pre)
    assert set(rd.keys()) == set(tensors)
    for k, v in tensors.items():
        if isinstance(v, bytes):
            assert rd.string_scalar(k) == v
        else:
            got = rd.tensor(k, verify=True)
            assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v)
    # a flipped byte in the index is caught by the block checksum
    bad = bytearray(raw)
    bad[10] ^= 0x40
    open(pre + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        tc.BundleReader(pre)


@pytest.mark.parametrize("conditional", [True, False])
def test_unet_checkpoint_roundtrip_is_independent_of_layer_numbering(tmp_path, conditional):
    """save -> load returns every parameter bit for bit although (a) the layer_with_weights-N numbers are shuffled (Keras
    assigns them by graph depth, not by creation) and (b) the per-class creation counters start at arbitrary offsets (a VQ-VAE
    built earlier in the process consumes names first): the mapping rests on creation order within each class only."""
    import dm3d_amd
    from dm3d_amd import tf_checkpoint as tc
    cfg = dm3d_amd.UNetConfig(img_size=4, img_channels=4, widths=(16, 32), has_attention=(False, True), num_res_blocks=1,
                              conditional=conditional, first_conv_channels=16)
    state = dm3d_amd.synthetic_weights(cfg, seed=11)
    pre = str(tmp_path / "unet")
    tc.save_unet_checkpoint(pre, state, cfg, first_index={"conv3d": 23, "dense": 1, "batch_normalization": 9}, shuffle_seed=5)
    got = tc.load_unet_state(pre, cfg, verify=True)
    assert set(got) == set(state)
    for k in state:
        assert np.array_equal(got[k], state[k]), k
    # same-shape layers must not be swapped: conv1 / conv2 of a width-preserving block differ in the fixture
    blocks = [k for k in state if k.endswith(".conv2.kernel") and state[k].shape == state[k.replace("conv2", "conv1")].shape]
    assert blocks and all(not np.array_equal(got[k], got[k.replace("conv2", "conv1")]) for k in blocks)
    # a different architecture is refused with a count message rather than loaded wrongly
    other = dm3d_amd.UNetConfig(img_size=4, img_channels=4, widths=(16, 32), has_attention=(False, True), num_res_blocks=2,
                                conditional=conditional, first_conv_channels=16)
    with pytest.raises(ValueError, match="layers"):
        tc.load_unet_state(pre, other)
    with pytest.raises(ValueError, match="not found"):
        tc.load_unet_state(pre, cfg, root=("ema_network",))


def test_keras_layer_plan_counts_match_the_reference_graph():
    """build_model at the BASELINE configuration creates 52 top-level Conv3D, 25 Dense, 35 BatchNormalization, 1 Embedding and
    6 CrossAttentionBlock layers with weights (conditional_dm3d.py:348-414; SURVEY appendix A block table)."""
    import collections
    import dm3d_amd
    from dm3d_amd import tf_checkpoint as tc
    cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
    n = collections.Counter(c for c, _ in tc.keras_layer_plan(cfg))
    # 17 ResidualBlocks: 2 k3 convs each + 12 k1 skips (3 widening down blocks, all 9 concat blocks); conv_in, conv_out,
    # 2 DownSample, 2 UpSample
    assert n["conv3d"] == 17 * 2 + 12 + 2 + 2 + 2
    assert n["dense"] == 2 + 17 + 6                          # TimeMLP, one temb Dense per ResidualBlock, one ContextMLP per attention
    assert n["batch_normalization"] == 17 * 2 + 1
    assert n["embedding"] == 1 and n["cross_attention_block"] == 6


def test_vqvae_checkpoint_roundtrip(tmp_path):
    import dm3d_amd  # noqa: F401
    from dm3d_amd import tf_checkpoint as tc
    from dm3d_amd.networks.vqvae3d_monai import vqvae_param_spec, keras_init_vqvae_weights
    spec = vqvae_param_spec(1, 1, (8, 16), 2, (8, 16), 32, 4, 16)
    state = keras_init_vqvae_weights(spec, seed=2)
    rng = np.random.default_rng(3)
    state = {k: (v + rng.normal(size=v.shape).astype(np.float32) * 0.1) for k, v in state.items()}     # make same-shape layers differ
    for root in ((), ("vqvae_trainer",)):
        pre = str(tmp_path / ("vq" + "_".join(root)))
        tc.save_vqvae_checkpoint(pre, state, spec, root=root, first_index={"conv3d": 4, "p_re_lu": 2})
        got = tc.load_vqvae_state(pre, spec, root=root, verify=True)
        assert set(got) == set(spec)
        for k in spec:
            assert np.array_equal(got[k], state[k]), k
    other = vqvae_param_spec(1, 1, (8, 16), 3, (8, 16), 32, 4, 16)
    with pytest.raises(ValueError, match="layers"):
        tc.load_vqvae_state(pre, other, root=("vqvae_trainer",))


# ---- round 5: fixtures assembled byte by byte from the published formats (tests/golden/make_tf_fixtures.py), NOT through tf_checkpoint.py's writer
GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def _expected(key, shape, dtype=np.float32):
    import zlib
    n = int(np.prod(shape)) if shape else 1
    return (np.arange(n, dtype=np.float64) * 0.25 + zlib.crc32(key.encode()) % 97).astype(dtype).reshape(shape)


def test_reads_hand_assembled_multi_shard_bundle(tmp_path):
    """What TensorFlow emits and this package's writer does not: three data shards with non-zero first offsets, an index of several data
    blocks with restart interval 16 and long shared key prefixes, shortest-separator index keys, a string tensor in a middle shard."""
    import os
    import shutil
    from dm3d_amd import tf_checkpoint as tc
    pre = os.path.join(GOLDEN, "tf_multi")
    rd = tc.BundleReader(pre)
    assert rd.num_shards == 3 and len(rd.keys()) == 63
    assert {rd.entries[k]["shard"] for k in rd.keys()} == {0, 1, 2}
    # the index really has several data blocks and prefix-compressed entries (else the fixture tests nothing)
    raw = open(pre + ".index", "rb").read()
    blocks = 0
    with open(pre + ".index", "rb") as f:
        footer = raw[-48:]
        pos = 0
        _, pos = tc._read_varint(footer, pos)
        _, pos = tc._read_varint(footer, pos)
        ioff, pos = tc._read_varint(footer, pos)
        isz, pos = tc._read_varint(footer, pos)
        for sep, handle in tc._block_entries(tc._read_block(f, ioff, isz, True)):
            blocks += 1
            boff, p2 = tc._read_varint(handle, 0)
            bsz, _ = tc._read_varint(handle, p2)
            body = tc._read_block(f, boff, bsz, True)
            shared_max = 0
            p = 0
            end = len(body) - 4 * (int.from_bytes(body[-4:], "little") + 1)
            while p < end:
                sh, p = tc._read_varint(body, p)
                un, p = tc._read_varint(body, p)
                vl, p = tc._read_varint(body, p)
                p += un + vl
                shared_max = max(shared_max, sh)
            assert shared_max >= 20 and int.from_bytes(body[-4:], "little") >= 1
    assert blocks == 4
    for k in rd.keys():
        e = rd.entries[k]
        if e["dtype"] == tc.DT_STRING:
            assert rd.string_scalar(k) == b"object graph bytes \x00\x01\xff stand-in"
            continue
        want = _expected(k, e["shape"], np.int64 if e["dtype"] == tc.DT_INT64 else np.float32)
        got = rd.tensor(k, verify=True)
        assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), k
    # a flipped payload byte is caught by the entry's checksum; a missing shard by name
    for suffix in (".index", ".data-00000-of-00003", ".data-00001-of-00003", ".data-00002-of-00003"):
        shutil.copy(pre + suffix, tmp_path / ("c" + suffix))
    bad = bytearray(open(tmp_path / "c.data-00002-of-00003", "rb").read())
    bad[40] ^= 1
    open(tmp_path / "c.data-00002-of-00003", "wb").write(bytes(bad))
    rb = tc.BundleReader(str(tmp_path / "c"))
    with pytest.raises(ValueError, match="checksum"):
        for k in rb.keys():
            if rb.entries[k]["shard"] == 2 and rb.entries[k]["dtype"] != tc.DT_STRING:
                rb.tensor(k, verify=True)
    os.remove(tmp_path / "c.data-00001-of-00003")
    with pytest.raises(FileNotFoundError):
        tc.BundleReader(str(tmp_path / "c")).string_scalar("_CHECKPOINTABLE_OBJECT_GRAPH")


def test_reads_partitioned_variables():
    """A variable saved in slices (tf partitioners / SaveSliceInfo): the full-tensor entry lists TensorSliceProtos, the pieces sit under
    checkpoint::EncodeTensorNameSlice keys (OrderedCode).  Rows in three pieces, columns in two."""
    import os
    from dm3d_amd import tf_checkpoint as tc
    assert [tc._oc_signed_increasing(v).hex() for v in (0, -1, 63, 64, 512, 8191, 8192, -64, -65)] == \
        ["80", "7f", "bf", "c040", "c200", "dfff", "e02000", "40", "3fbf"]              # ordered_code.cc's own examples
    assert tc.encode_slice_key("a", [(0, -1), (4, 2)]) == b"\x00" + b"a\x00\x01" + b"\x01\x02" + b"\x7f\x7f" + b"\x84\x82"
    rd = tc.BundleReader(os.path.join(GOLDEN, "tf_sliced"))
    assert rd.keys() == ["dense/bias", "dense/kernel", "embedding/table"]
    assert rd.entries["embedding/table"]["slices"] == [((0, 4), (0, -1)), ((4, 4), (0, -1)), ((8, 2), (0, -1))]
    for k in rd.keys():
        got = rd.tensor(k, verify=True)
        assert got.shape == rd.shape(k) and np.array_equal(got, _expected(k, rd.shape(k))), k


def test_fixture_generator_is_independent_and_reproducible(tmp_path, monkeypatch):
    """The committed bytes are what the generator writes (the files were not edited by hand or by another tool), and the generator does
    not import the package."""
    import importlib.util
    import os
    src = open(os.path.join(GOLDEN, "make_tf_fixtures.py")).read()
    assert "import dm3d_amd" not in src and "tf_checkpoint import" not in src and "from dm3d_amd" not in src
    spec = importlib.util.spec_from_file_location("mk_fix", os.path.join(GOLDEN, "make_tf_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    monkeypatch.setattr(mk, "HERE", str(tmp_path))
    mk.multi_shard_fixture()
    mk.sliced_fixture()
    for f in sorted(os.listdir(tmp_path)):
        assert open(tmp_path / f, "rb").read() == open(os.path.join(GOLDEN, f), "rb").read(), f
    from dm3d_amd import tf_checkpoint as tc
    assert mk.crc32c_bitwise(b"123456789") == tc.crc32c(b"123456789") == 0xE3069283


def test_vqgan_checkpoint_mapper_skips_the_gan_half(tmp_path):
    """networks/vqgan.py checkpoints (vqgan.py:599-703) hold two discriminators, LPIPS and optimizer state beside the autoencoder; their
    Conv3D layers are numbered AFTER the autoencoder's.  Only encoder / decoder / quantizer are read."""
    from dm3d_amd import tf_checkpoint as tc
    from dm3d_amd.networks.vqgan import vqgan_param_spec
    spec = vqgan_param_spec(2, 2, (8, 16), 1, (8, 16), 32, 8, 16)
    rng = np.random.default_rng(3)
    state = {k: rng.normal(size=v).astype(np.float32) for k, v in spec.items()}
    extra = {"discriminator/conv3d_40/kernel": rng.normal(size=(4, 4, 4, 1, 8)).astype(np.float32), "discriminator/conv3d_40/bias": np.zeros(8, np.float32),
             "discriminator_2d/dense_3/kernel": np.ones((5, 1), np.float32)}
    pre = str(tmp_path / "gan")
    tc.save_vqvae_checkpoint(pre, state, spec, extra=extra)
    got = tc.load_vqvae_state(pre, spec, parts=("encoder", "decoder", "quantizer"))
    assert set(got) == set(spec) and all(np.array_equal(got[k], state[k]) for k in spec)
    with pytest.raises(ValueError, match="conv3d"):
        tc.load_vqvae_state(pre, spec)                    # reading everything counts the discriminator's convs too
    with pytest.raises(ValueError, match="quantizer"):
        tc.load_vqvae_state(pre, spec, root=("encoder",), parts=("encoder", "decoder", "quantizer"))


def test_unet_checkpoint_carries_adam_slots(tmp_path):
    """model.load_weights(<epoch>.ckpt) of a compiled model resumes Adam (main_conditional_dm.py:174-183): optimizer/iter and the m / v
    slot variables travel in the OptimizerV2 layout (slot_variables of the optimizer object, <variable>/.OPTIMIZER_SLOT/optimizer/<slot> keys)."""
    import dm3d_amd
    from dm3d_amd import tf_checkpoint as tc
    from dm3d_amd.train import is_trainable
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    W = dm3d_amd.synthetic_weights(cfg, seed=1)
    rng = np.random.default_rng(4)
    opt = {"optimizer/iter": np.asarray(17, np.int64)}
    for n, v in W.items():
        if is_trainable(n):
            opt[f"optimizer/m/{n}"] = rng.normal(size=v.shape).astype(np.float32)
            opt[f"optimizer/v/{n}"] = rng.random(size=v.shape).astype(np.float32)
    pre = str(tmp_path / "ep")
    tc.save_unet_checkpoint(pre, W, cfg, shuffle_seed=5, optimizer=opt)
    rd = tc.BundleReader(pre)
    assert any("/.OPTIMIZER_SLOT/optimizer/m/" in k for k in rd.keys()) and "optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE" in rd.keys()
    back = tc.load_unet_state(pre, cfg, with_optimizer=True)
    assert int(back["optimizer/iter"]) == 17
    assert set(back) == set(W) | set(opt)
    assert all(np.array_equal(back[k], opt[k]) for k in opt) and all(np.array_equal(back[k], W[k]) for k in W)
    assert set(tc.load_unet_state(pre, cfg)) == set(W)            # the weights alone, as before
