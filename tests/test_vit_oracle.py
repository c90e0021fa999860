"""Row A1 oracle check (CPU): the plain-PyTorch ViT restatement in oracle/oracle.py equals the
independent `transformers.Dinov2Model` implementation on the same seeded weights.  The reference's
own model (torch.hub FeatUp -> facebookresearch/dinov2) cannot be fetched offline: parity unpinned."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from vfmreg import vit as V


def _hf_model(w, dim, depth, mlp):
    tr = pytest.importorskip("transformers")
    cfg = tr.Dinov2Config(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=dim // 64, patch_size=14,
                          image_size=518, mlp_ratio=mlp // dim, layer_norm_eps=1e-6, layerscale_value=1.0,
                          qkv_bias=True, hidden_act="gelu")
    m = tr.Dinov2Model(cfg).eval()
    sd = m.state_dict()
    t = lambda a: torch.from_numpy(np.asarray(a))
    new = {"embeddings.cls_token": t(w["cls_token"]), "embeddings.position_embeddings": t(w["pos_embed"]),
           "embeddings.patch_embeddings.projection.weight": t(w["patch_embed.proj.weight"]),
           "embeddings.patch_embeddings.projection.bias": t(w["patch_embed.proj.bias"]),
           "layernorm.weight": t(w["norm.weight"]), "layernorm.bias": t(w["norm.bias"])}
    for i in range(depth):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        qkv_w, qkv_b = w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"]
        for j, name in enumerate(("query", "key", "value")):
            new[q + f"attention.attention.{name}.weight"] = t(qkv_w[j * dim:(j + 1) * dim])
            new[q + f"attention.attention.{name}.bias"] = t(qkv_b[j * dim:(j + 1) * dim])
        new[q + "attention.output.dense.weight"] = t(w[p + "attn.proj.weight"])
        new[q + "attention.output.dense.bias"] = t(w[p + "attn.proj.bias"])
        new[q + "layer_scale1.lambda1"] = t(w[p + "ls1.gamma"])
        new[q + "layer_scale2.lambda1"] = t(w[p + "ls2.gamma"])
        for a, b in (("norm1", "norm1"), ("norm2", "norm2"), ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2")):
            new[q + b + ".weight"] = t(w[p + a + ".weight"])
            new[q + b + ".bias"] = t(w[p + a + ".bias"])
    missing = [k for k in sd if k not in new and "mask_token" not in k]
    assert not missing, missing
    m.load_state_dict({**{k: v for k, v in sd.items() if "mask_token" in k}, **new})
    return m


def test_vit_restatement_equals_transformers_dinov2():
    dim, depth, mlp = 128, 3, 512
    w = V.random_weights(seed=3, dim=dim, depth=depth, mlp=mlp)
    rng = np.random.default_rng(0)
    # 518 x 518 with 37 patch rows: no resize and no position-embedding interpolation in either model
    img = rng.integers(0, 256, (1, 518, 518, 3), dtype=np.uint8)
    feats = orc.vit_reference(w, img, patch_h=37)  # [1, 37, 37, dim], includes ChannelNorm
    m = _hf_model(w, dim, depth, mlp)
    x = torch.from_numpy(img).permute(0, 3, 1, 2).float() / 255.0
    x = (x - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    with torch.no_grad():
        tok = m(pixel_values=x).last_hidden_state[:, 1:]
        ref = torch.nn.functional.layer_norm(tok, (dim,), torch.from_numpy(w["channel_norm.weight"]),
                                             torch.from_numpy(w["channel_norm.bias"]), 1e-5)
    np.testing.assert_allclose(feats.reshape(1, -1, dim), ref.numpy(), rtol=0, atol=2e-4)


def test_pos_embed_interpolation_and_patch_width():
    w = V.random_weights(seed=1, dim=64, depth=1, mlp=128)
    pe = V.interpolate_pos_embed(w["pos_embed"], 16, 21)
    assert pe.shape == (1 + 16 * 21, 64)
    np.testing.assert_array_equal(pe[0], w["pos_embed"][0, 0])
    np.testing.assert_allclose(pe, orc.interpolate_pos_embed(w["pos_embed"], 16, 21)[0].numpy(), atol=0)
    # image_features.py:68-69: 1200 x 1600 -> 16 x 21 patches (224 x 294), NCLT 700 x 820 -> 16 x 18
    assert int((224 / 1200) * 1600 / 14) == 21 and int((224 / 700) * 820 / 14) == 18


def test_fragment_tiling_roundtrip():
    rng = np.random.default_rng(0)
    W = rng.standard_normal((70, 50)).astype(np.float32)
    F = V.to_frag_f16(W)
    assert F.shape == (3, 4, 2, 32, 8)
    for (n, k) in ((0, 0), (33, 17), (69, 49), (31, 15)):
        assert F[n // 32, k // 16, (k // 8) % 2, n % 32, k % 8] == np.float16(W[n, k])
    assert F[2, 3, 1, 31, 7] == 0  # padding


def test_load_state_dict_from_three_checkpoint_layouts():
    """vit.load_state_dict (the loader image_features.py:39-44 needs offline): a transformers.Dinov2Model state dict
    (seeded weights), the same weights in facebookresearch/dinov2 naming (flat and chunked blocks) and inside the FeatUp
    wrapper's naming all convert to the same arrays, and the converted weights reproduce the HF model's tokens."""
    dim, depth, mlp = 128, 3, 512
    w = V.random_weights(seed=5, dim=dim, depth=depth, mlp=mlp)
    m = _hf_model(w, dim, depth, mlp)
    hf = V.load_state_dict(m.state_dict())                       # Hugging Face naming, no ChannelNorm -> identity
    for k, v in w.items():
        if k.startswith("channel_norm"):
            continue
        np.testing.assert_array_equal(hf[k], v, err_msg=k)
    assert (hf["channel_norm.weight"] == 1).all() and (hf["channel_norm.bias"] == 0).all()
    with pytest.raises(KeyError, match="ChannelNorm"):
        V.load_state_dict(m.state_dict(), channel_norm="require")
    # facebookresearch/dinov2 naming (+ mask_token, chunked blocks) and the FeatUp wrapper around it
    fb = {k: torch.from_numpy(v) for k, v in w.items() if not k.startswith("channel_norm")}
    fb["mask_token"] = torch.zeros(1, dim)
    chunked = {(k.replace("blocks.", "blocks.0.", 1) if k.startswith("blocks.") else k): v for k, v in fb.items()}
    featup = {"model.0.model." + k: v for k, v in fb.items()}
    featup["model.1.norm.weight"] = torch.from_numpy(w["channel_norm.weight"])
    featup["model.1.norm.bias"] = torch.from_numpy(w["channel_norm.bias"])
    featup["upsampler.up1.range_temp"] = torch.zeros(1)
    for sd, has_cn in ((fb, False), (chunked, False), (featup, True)):
        got = V.load_state_dict(sd)
        assert set(got) == set(w)
        for k, v in w.items():
            if k.startswith("channel_norm") and not has_cn:
                continue
            np.testing.assert_array_equal(got[k], v, err_msg=k)
    # the converted HF checkpoint drives the oracle ViT to the HF model's own output
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (1, 518, 518, 3), dtype=np.uint8)
    feats = orc.vit_reference(hf, img, patch_h=37)               # identity ChannelNorm = plain LayerNorm over channels
    x = torch.from_numpy(img).permute(0, 3, 1, 2).float() / 255.0
    x = (x - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    with torch.no_grad():
        tok = m(pixel_values=x).last_hidden_state[:, 1:]
        ref = torch.nn.functional.layer_norm(tok, (dim,), None, None, 1e-5)
    np.testing.assert_allclose(feats.reshape(1, -1, dim), ref.numpy(), rtol=0, atol=2e-4)
    # malformed inputs are loud
    bad = dict(fb)
    del bad["blocks.1.attn.proj.bias"]
    with pytest.raises(KeyError, match="lacks"):
        V.load_state_dict(bad)
    with pytest.raises(NotImplementedError, match="register"):
        V.load_state_dict({**fb, "register_tokens": torch.zeros(1, 4, dim)})
