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
