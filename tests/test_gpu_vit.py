"""Row A1 on the GPU: the HIP ViT-S/14 forward (fp16 MFMA, fp32 accumulate / residual) against the
plain PyTorch fp32 reference of the same op (oracle.vit_reference) on identical seeded weights and
images.  Floating point => tolerance parity: max |err| <= 1e-2 on O(1) ChannelNorm outputs and
per-token cosine >= 0.99999 (measured 2.8e-3 / 0.9999997; fp16 operand rounding through 12 blocks; the reference itself runs fp32)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _smooth_images(rng, B, H, W):
    low = rng.uniform(0, 255, (B, H // 40 + 2, W // 40 + 2, 3)).astype(np.float32)
    t = torch.from_numpy(low).permute(0, 3, 1, 2)
    up = torch.nn.functional.interpolate(t, size=(H, W), mode="bilinear", align_corners=False)
    up = up + 12.0 * torch.randn(up.shape, generator=torch.Generator().manual_seed(1))
    return up.clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8).contiguous().numpy()


def _check(w, imgs, atol, cos_min):
    from oracle import oracle as orc
    from vfmreg import vit as V
    B, H, W, _ = imgs.shape
    model = V.ViTS14(w, H, W, device="cuda")
    out = model.forward(torch.from_numpy(imgs).cuda())
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    ref = orc.vit_reference(w, imgs)
    assert out.shape == ref.shape
    err = np.abs(out - ref)
    cos = (out * ref).sum(-1) / (np.linalg.norm(out, axis=-1) * np.linalg.norm(ref, axis=-1))
    assert np.isfinite(out).all()
    assert cos.min() >= cos_min, f"min token cosine {cos.min()}, max abs err {err.max()}"
    assert err.max() <= atol, f"max abs err {err.max()} (mean {err.mean()})"
    return err.max(), cos.min()


def test_vit_with_dinov2_like_statistics():
    """VERDICT r4 item 7b / ADVICE r4: a trained DINOv2 carries a few residual channels 50-200x the rest and LayerScale gammas over
    four decades; seeded random weights have neither.  `dinov2_like_weights` plants both (6 channels at 50-200x on the residual path
    of every token, gamma log-uniform in [1e-4, 1]); the fp16-operand forward with the LayerNorm folded into the GEMMs must hold the
    same tolerances against the fp32 oracle -- on the full output AND on the channels that are NOT outliers (the planted channels
    alone would carry a cosine to 1), at C3's resolution with all 12 blocks."""
    from oracle import oracle as orc
    from vfmreg import vit as V
    w, ch = V.dinov2_like_weights(seed=21)
    imgs = _smooth_images(np.random.default_rng(8), 6, 1200, 1600)
    model = V.ViTS14(w, 1200, 1600, device="cuda")
    out = model.forward(torch.from_numpy(imgs).cuda()).cpu().numpy()
    ref = orc.vit_reference(w, imgs)
    assert np.isfinite(out).all()
    rest = np.setdiff1d(np.arange(384), ch)
    cos = lambda a, b: (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))   # noqa: E731
    c_all, c_rest = cos(out, ref).min(), cos(out[..., rest], ref[..., rest]).min()
    err = np.abs(out - ref).max()
    rel_rest = np.abs(out[..., rest] - ref[..., rest]).max() / np.abs(ref[..., rest]).max()
    print("dinov2-like: max abs err", err, "min cosine", c_all, "min cosine off the outlier channels", c_rest, "rel err there", rel_rest,
          "| ref abs max", np.abs(ref).max(), "off-outlier abs max", np.abs(ref[..., rest]).max())
    assert c_all >= 0.99999 and c_rest >= 0.99999, (c_all, c_rest)
    assert err <= 1e-2 and rel_rest <= 1e-2, (err, rel_rest)


def test_vit_rows_with_a_large_common_offset():
    """tokens whose |mean| is several times their spread -- the case where a LayerNorm computed from an fp16 copy of the
    un-normalised row loses bits to the subtraction.  A common offset of 4x the spread on every channel of the residual path."""
    from vfmreg import vit as V
    w = V.random_weights(seed=23, dim=384, depth=4, mlp=1536)
    w["patch_embed.proj.bias"] += 4.0
    w["cls_token"] += 4.0
    imgs = _smooth_images(np.random.default_rng(9), 2, 560, 700)
    _check(w, imgs, atol=1e-2, cos_min=0.99999)


def test_vit_small_config():
    from vfmreg import vit as V
    w = V.random_weights(seed=5, dim=128, depth=2, mlp=256)
    imgs = _smooth_images(np.random.default_rng(0), 2, 300, 400)
    _check(w, imgs, atol=1e-2, cos_min=0.99999)


def test_vit_s14_six_cameras_1600x1200():
    """BASELINE config C3's feature stage: 6 x 1200 x 1600 surround images -> 6 x 16 x 21 x 384."""
    from vfmreg import vit as V
    w = V.random_weights(seed=0)
    imgs = _smooth_images(np.random.default_rng(1), 6, 1200, 1600)
    e, c = _check(w, imgs, atol=1e-2, cos_min=0.99999)
    print("vit-s/14 6 cams: max abs err", e, "min cosine", c)


def test_vit_s14_through_the_one_workgroup_qkv_attention_kernel_against_the_fp32_reference():
    """The same comparison with `vit_fused_qkv` = 1: QKV + attention of every (image, head) in vit_qkv_attention_kernel (what batches of
    24 images and more take by policy), straight against the fp32 oracle -- 12 blocks at the 6-camera size, and a token count that is
    no multiple of 32 (700 x 820: 289 tokens, keys masked in the last tile)."""
    from vfmreg import _lib
    from vfmreg import vit as V
    with _lib.using(_lib.Config().set("vit_fused_qkv", 1).set("vit_fused_mlp", 1)):   # (and fc1 -> GELU -> fc2 in vit_mlp_kernel)
        _check(V.random_weights(seed=0), _smooth_images(np.random.default_rng(1), 3, 1200, 1600), atol=1e-2, cos_min=0.99999)
        _check(V.random_weights(seed=2, dim=384, depth=3, mlp=1536), _smooth_images(np.random.default_rng(2), 2, 700, 820), atol=1e-2, cos_min=0.99999)


def test_vit_nclt_resolution_padding():
    """700 x 820 -> 16 x 18 patches = 289 tokens: not a multiple of 32 (key masking / padded rows)."""
    from vfmreg import vit as V
    w = V.random_weights(seed=2, dim=384, depth=3, mlp=1536)
    imgs = _smooth_images(np.random.default_rng(2), 1, 700, 820)
    _check(w, imgs, atol=1e-2, cos_min=0.99999)


def test_vit_b14_config_c5():
    """BASELINE config C5's feature extractor: ViT-B/14 (dim 768, 12 heads, MLP 3072) -> 768-D descriptors.
    4 blocks keep the CPU reference within seconds; every kernel shape of the 12-block model is exercised."""
    from vfmreg import vit as V
    w = V.random_weights(seed=3, dim=768, depth=4, mlp=3072)
    imgs = _smooth_images(np.random.default_rng(3), 2, 600, 800)
    _check(w, imgs, atol=1e-2, cos_min=0.99999)


def test_vit_lds_tiled_gemms_equal_the_direct_kernels_bit_for_bit():
    """The LDS-staged 128 x 128 workgroup-tile GEMM (round 4: chosen for batches of many images) accumulates every output over the same
    k order with the same instruction as the direct kernel, so the two forwards are identical bits -- also where the token tiles do
    not fill the last group of four (5 images x 288 padded tokens = 45 tiles), at ViT-S and at a ViT-B-like width, with the residual
    GEMMs as 128 x 128 tiles and as one 128 x 384 tile per workgroup; and the forced LDS path passes the oracle tolerance."""
    from vfmreg import _lib
    from vfmreg import vit as V
    lib = _lib.load()
    try:
        for (dim, depth, mlp, B, H, W) in ((384, 2, 1536, 5, 560, 700), (768, 1, 3072, 3, 560, 700), (384, 12, 1536, 6, 1200, 1600)):
            w = V.random_weights(seed=11, dim=dim, depth=depth, mlp=mlp)
            imgs = torch.from_numpy(_smooth_images(np.random.default_rng(3), B, H, W)).cuda()
            model = V.ViTS14(w, H, W, device="cuda")
            lib.vfm_debug_set_vit_gemm(-5, 0)        # never the LDS kernel
            direct = model.forward(imgs).clone()
            lib.vfm_debug_set_vit_gemm(-5, 1)        # always (every GEMM whose N is a multiple of 128)
            for wide in (0, 1):                      # the residual GEMMs (N = 384) as 128 x 128 tiles / as one 128 x 384 tile per workgroup (round 5)
                lib.vfm_debug_set_vit_gemm(-17, wide)
                tiled = model.forward(imgs).clone()
                torch.cuda.synchronize()
                assert torch.equal(direct, tiled), (dim, B, wide, float((direct - tiled).abs().max()))
        lib.vfm_debug_set_vit_gemm(-5, 1)
        w = V.random_weights(seed=5, dim=128, depth=2, mlp=256)
        _check(w, _smooth_images(np.random.default_rng(0), 9, 300, 400), atol=1e-2, cos_min=0.99999)
    finally:
        lib.vfm_debug_set_vit_gemm(-5, 256)
        lib.vfm_debug_set_vit_gemm(-17, 0)


def test_vit_token_stationary_gemms_equal_the_other_kernels_bit_for_bit():
    """vit_gemm_astat_kernel (end of round 4: the QKV / fc1 products of large batches with 128 token rows resident in the LDS, the weight
    fragments streamed through registers by waves that share nothing else) accumulates every output over the same k order with the same
    instruction as the direct and the LDS-tiled kernels: identical bits -- also where the token tiles do not fill the last group of four,
    and at a ViT-B-like width, whose rows (192 KiB) do not fit the LDS: the launcher then keeps the other kernels."""
    from vfmreg import _lib
    from vfmreg import vit as V
    lib = _lib.load()
    try:
        for (dim, depth, mlp, B, H, W) in ((384, 2, 1536, 5, 560, 700), (384, 1, 1536, 7, 1200, 1600), (768, 1, 3072, 3, 560, 700),
                                           (384, 12, 1536, 6, 1200, 1600)):
            w = V.random_weights(seed=17, dim=dim, depth=depth, mlp=mlp)
            imgs = torch.from_numpy(_smooth_images(np.random.default_rng(5), B, H, W)).cuda()
            model = V.ViTS14(w, H, W, device="cuda")
            lib.vfm_debug_set_vit_gemm(-9, -1)       # never
            ref = model.forward(imgs).clone()
            lib.vfm_debug_set_vit_gemm(-9, 1)        # always (QKV and fc1, wherever 128 rows of A fit the LDS)
            for two in (0, 1):                       # round 4's form (one channel tile per wave) and round 5's (two: half the LDS reads per MFMA)
                lib.vfm_debug_set_vit_gemm(-15, two)
                got = model.forward(imgs).clone()
                torch.cuda.synchronize()
                assert torch.equal(ref, got), (dim, B, two, float((ref - got).abs().max()))
    finally:
        lib.vfm_debug_set_vit_gemm(-15, 1)
        lib.vfm_debug_set_vit_gemm(-9, 0)            # the default policy: where its rounds of one workgroup per compute unit are full


def test_vit_attention_with_keys_and_values_in_the_lds_equals_the_per_tile_kernel_bit_for_bit():
    """vit_attention_lds_kernel (K / V^T of an (image, head) staged once per workgroup, four query tiles per workgroup) against the
    one-wave-per-tile kernel: the same MFMAs over the same fragments in the same order -- identical bits; token counts whose last group
    has one, two or three query tiles (9, 10, 11 tiles), one image and several."""
    from vfmreg import _lib
    from vfmreg import vit as V
    lib = _lib.load()
    try:
        for (dim, depth, mlp, B, H, W) in ((384, 2, 1536, 5, 560, 700), (384, 1, 1536, 1, 1200, 1600), (768, 1, 3072, 2, 560, 600),
                                           (384, 12, 1536, 6, 1200, 1600)):
            w = V.random_weights(seed=13, dim=dim, depth=depth, mlp=mlp)
            imgs = torch.from_numpy(_smooth_images(np.random.default_rng(4), B, H, W)).cuda()
            model = V.ViTS14(w, H, W, device="cuda")
            lib.vfm_debug_set_vit_gemm(-7, 0)        # never
            per_tile = model.forward(imgs).clone()
            lib.vfm_debug_set_vit_gemm(-7, 1)        # always
            shared = model.forward(imgs).clone()
            torch.cuda.synchronize()
            assert torch.equal(per_tile, shared), (dim, B, H, W, float((per_tile - shared).abs().max()))
    finally:
        lib.vfm_debug_set_vit_gemm(-7, 1)


def test_vit_qkv_and_attention_in_one_workgroup_equal_the_two_kernels_bit_for_bit():
    """vit_qkv_attention_kernel (round 6: q, K and V^T of an (image, head) never leave the compute unit) against the QKV GEMM + the attention
    kernel: the same MFMAs over the same fragments in the same k order and the same epilogue arithmetic -- identical bits.  Token counts
    with 1, 4, 6, 9, 10, 11 and 12 tiles (the waves' last tiles missing or not, one to three tiles per wave), image counts that are no multiple of 8 (the XCD mapping's empty
    slots), one layer and twelve; a width the kernel is not built for (768) takes the two kernels whatever the key says."""
    from vfmreg import _lib
    from vfmreg import vit as V
    for (dim, depth, mlp, B, H, W) in ((384, 2, 1536, 5, 560, 700), (384, 1, 1536, 1, 1200, 1600), (384, 1, 1536, 9, 1200, 1800),
                                       (384, 1, 1536, 3, 300, 200), (768, 1, 3072, 2, 560, 600), (384, 12, 1536, 13, 1200, 1600),
                                       # 12, 9, 10, 4 and 1 token tiles (1200 x 1800 above has 13: the two kernels either way)
                                       (384, 1, 1536, 7, 1200, 1700), (384, 1, 1536, 4, 1200, 1300), (384, 1, 1536, 2, 1200, 1400),
                                       (384, 1, 1536, 10, 1200, 500), (384, 2, 1536, 17, 1200, 100)):
        w = V.random_weights(seed=13, dim=dim, depth=depth, mlp=mlp)
        imgs = torch.from_numpy(_smooth_images(np.random.default_rng(4), B, H, W)).cuda()
        model = V.ViTS14(w, H, W, device="cuda")
        with _lib.using(_lib.Config().set("vit_fused_qkv", -1).set("vit_fused_mlp", -1)):
            two = model.forward(imgs).clone()
        with _lib.using(_lib.Config().set("vit_fused_qkv", 1)):
            one = model.forward(imgs).clone()
        torch.cuda.synchronize()
        assert torch.isfinite(one).all()
        assert torch.equal(two, one), (dim, B, H, W, float((two - one).abs().max()))


def test_vit_mlp_in_one_workgroup_equals_fc1_and_fc2_bit_for_bit():
    """vit_mlp_kernel (round 6: fc1 -> GELU -> fc2 of 128 tokens in one workgroup, the hidden activations never leave the registers) against
    the two GEMM kernels: the same MFMAs over the same fragments in the same k order, the same epilogues -- identical bits.  Token counts
    whose last group of four tiles is partial, one layer and twelve, with and without the one-workgroup QKV + attention in front; a width it
    is not built for (768) takes the two kernels whatever the key says."""
    from vfmreg import _lib
    from vfmreg import vit as V
    for (dim, depth, mlp, B, H, W) in ((384, 2, 1536, 5, 560, 700), (384, 1, 1536, 1, 1200, 1600), (384, 1, 1536, 3, 300, 200),
                                       (768, 1, 3072, 2, 560, 600), (384, 12, 1536, 13, 1200, 1600), (384, 1, 1536, 7, 1200, 1700)):
        w = V.random_weights(seed=13, dim=dim, depth=depth, mlp=mlp)
        imgs = torch.from_numpy(_smooth_images(np.random.default_rng(4), B, H, W)).cuda()
        model = V.ViTS14(w, H, W, device="cuda")
        with _lib.using(_lib.Config().set("vit_fused_mlp", -1).set("vit_fused_qkv", -1)):
            two = model.forward(imgs).clone()
        for qkv in (-1, 1):
            with _lib.using(_lib.Config().set("vit_fused_mlp", 1).set("vit_fused_qkv", qkv)):
                one = model.forward(imgs).clone()
            torch.cuda.synchronize()
            assert torch.isfinite(one).all()
            assert torch.equal(two, one), (dim, B, H, W, qkv, float((two - one).abs().max()))


def test_vit_preprocessing_one_workgroup_per_patch_equals_the_per_unit_kernel_bit_for_bit():
    """vit_preprocess_patch_kernel (round 5: a workgroup per 14 x 14 patch, each thread's two source rows read once for the three channels,
    the token's fragment units assembled in the LDS) evaluates the expressions of round 1's kernel in the same order: identical features --
    C3's resolution, NCLT's (padding tokens), an up-sampling resize (source smaller than 224 rows: taps at the right / bottom border)."""
    from vfmreg import _lib
    from vfmreg import vit as V
    lib = _lib.load()
    try:
        for (B, H, W) in ((6, 1200, 1600), (2, 700, 820), (3, 150, 210)):
            w = V.random_weights(seed=19, dim=384, depth=1, mlp=1536)
            imgs = torch.from_numpy(_smooth_images(np.random.default_rng(6), B, max(H, 80), max(W, 80))[:, :H, :W].copy()).cuda()
            model = V.ViTS14(w, H, W, device="cuda")
            lib.vfm_debug_set_vit_gemm(-14, 0)
            old = model.forward(imgs).clone()
            lib.vfm_debug_set_vit_gemm(-14, 1)
            new = model.forward(imgs).clone()
            torch.cuda.synchronize()
            assert torch.equal(old, new), (B, H, W, float((old - new).abs().max()))
    finally:
        lib.vfm_debug_set_vit_gemm(-14, 1)


def test_vit_batch_as_two_half_batches_on_two_streams_equals_one_forward_bit_for_bit():
    """ViTS14.SPLIT_FROM (opt-in, round 5): batches from that many images on run as two half-batches on two side streams, the second
    half enqueued by a helper thread (each kernel's tail under the other half's next kernel): the images are independent, so the outputs
    are those of one forward, bit for bit -- odd batch sizes, a caller's own stream, two calls in a row (the halves' workspaces are reused),
    and a batch right under the threshold."""
    from vfmreg import vit as V
    w = V.random_weights(seed=4, dim=384, depth=2, mlp=1536)
    H, W = 280, 350
    model = V.ViTS14(w, H, W, device="cuda")
    rng = np.random.default_rng(2)
    old = V.ViTS14.SPLIT_FROM
    try:
        for B in (63, 64, 71):
            imgs = torch.from_numpy(rng.integers(1, 255, (B, H, W, 3), dtype=np.uint8)).cuda()
            V.ViTS14.SPLIT_FROM = 64
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                a = model.forward(imgs).clone()
                b = model.forward(imgs).clone()
            st.synchronize()
            V.ViTS14.SPLIT_FROM = 0
            ref = model.forward(imgs).clone()
            torch.cuda.synchronize()
            assert torch.equal(a, ref) and torch.equal(b, ref), B
    finally:
        V.ViTS14.SPLIT_FROM = old
