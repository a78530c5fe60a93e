"""CPU: host-side facts of the latent path (no GPU, no compute through the library): the restatement in oracle/latent_oracle.py runs on
the product classes' state_dicts (so both readings of the reference's module tree agree on every key and shape), the documented
shapes and the decode-length behaviour, and the sampler's slice policy."""
import pytest
import torch

from oracle import latent_oracle as O


def small_inputs(B, D, L, T, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, D, L), generator=g)
    t = torch.randint(0, T, (B,), generator=g)
    cat = torch.cat([torch.randint(0, 256, (B, 4), generator=g), torch.randint(0, 33, (B, 1), generator=g)], dim=1)
    return x, t, cat, torch.randn((B, 3), generator=g)


def test_sdunet_car_keys_shapes_and_parameter_count():
    from trajectorydiffusion_amd import SDUNet_CAR
    m = SDUNet_CAR(traj_dim=4, channel_schedule=[128, 256, 512, 1024], diffusion_steps=500, res_blocks=4)
    assert sum(p.numel() for p in m.parameters()) == 162_682_884       # 162.7 M parameters
    sd = m.state_dict()
    assert len(sd) == 388
    # stage layouts of StableDiffusion.py:154-171: 4 blocks + stride-2 conv per encoder stage; 3 blocks, Upsample, 1 block per decoder stage
    assert sd["down_blocks.0.4.weight"].shape == (256, 256, 3) and "down_blocks.0.3.cond_proj.0.weight" in sd
    assert "up_blocks.0.3.stage1.0.weight" not in sd and sd["up_blocks.0.4.stage1.2.weight"].shape == (512, 512, 3)
    assert sd["up_blocks.0.0.shortcut.weight"].shape == (512, 2048, 1)
    assert sd["down_blocks.2.1.multihead_attn.in_proj_weight"].shape == (3072, 1024)
    assert sd["down_blocks.2.1.cond_proj.0.weight"].shape == (7 * 1024, 896)
    assert sd["embed_block.mix_fc.3.weight"].shape == (896, 896) and sd["head.1.weight"].shape == (4, 128, 3)
    assert "embed_block.pos_enc" not in sd                               # a plain attribute in the reference (StableDiffusionBlocks.py:95)


def test_oracle_runs_on_the_product_state_dict_small_model():
    from trajectorydiffusion_amd import SDUNet_CAR
    torch.manual_seed(0)
    m = SDUNet_CAR(4, [32, 64, 64], 50, res_blocks=3)                    # (the oracle is free of the kernels' channel limits)
    x, t, cat, num = small_inputs(3, 4, 12, 50)
    y = O.sdunet_car_forward(m.state_dict(), x, t, cat, num, [32, 64, 64], 50, 3)
    assert y.shape == (3, 4, 12) and torch.isfinite(y).all()
    # trajectories are independent: a batch permutation permutes the output
    perm = torch.tensor([2, 0, 1])
    y2 = O.sdunet_car_forward(m.state_dict(), x[perm], t[perm], cat[perm], num[perm], [32, 64, 64], 50, 3)
    assert torch.allclose(y2, y[perm], atol=1e-6)
    with pytest.raises(RuntimeError):                                    # 10 -> 5 -> odd: the decoder's torch.cat cannot line up (StableDiffusion.py:200)
        O.sdunet_car_forward(m.state_dict(), x[:, :, :10], t, cat, num, [32, 64, 64], 50, 3)


def test_condition_mix_never_calls_categorical_fc():
    """StableDiffusionBlocks.py:134-140: forward concatenates the gathered cell / day embeddings as they are; `categorical_fc` (:109) is
    a constructed-but-unused layer there (unlike TrajUNet's WideAndDeepEmbedBlock, TrajUNetBlocks.py:56).  The restatement must not
    depend on its weights, and the categorical slice it feeds mix_fc is LeakyReLU of the raw embedding rows."""
    from trajectorydiffusion_amd import ConditionMixBlock
    torch.manual_seed(3)
    m = ConditionMixBlock(50, hidden_dim=128, embed_dim=128)
    sd = {"embed_block." + k: v.detach().clone() for k, v in m.state_dict().items()}
    assert sd["embed_block.categorical_fc.weight"].shape == (640, 640)    # still a state_dict entry (strict loading, Utils.py:10-12)
    _, t, cat, num = small_inputs(5, 4, 8, 50, seed=4)
    a = O.condition_mix(sd, t, cat, num, 50)
    sd2 = dict(sd)
    sd2["embed_block.categorical_fc.weight"] = torch.randn(640, 640)
    sd2["embed_block.categorical_fc.bias"] = torch.randn(640)
    assert torch.equal(O.condition_mix(sd2, t, cat, num, 50), a)
    sd3 = dict(sd)
    sd3["embed_block.day_embedder.weight"] = sd["embed_block.day_embedder.weight"] + 0.5
    assert not torch.equal(O.condition_mix(sd3, t, cat, num, 50), a)


def test_vae_oracle_shapes_and_the_decode_length_behaviour():
    from trajectorydiffusion_amd import VAE_InvTrans
    torch.manual_seed(1)
    v = VAE_InvTrans().eval()
    sd = v.state_dict()
    assert len(sd) == 126
    z = O.vae_encode(sd, torch.randn(2, 2, 199))
    assert z.shape == (2, 4, 24)                                         # VAE.py:246: encode of a length-199 trajectory
    assert O.vae_decode(sd, z).shape == (2, 2, 24)                       # pooled to the LATENT length (VAE.py:191-192), not to 199
    assert O.vae_encode(sd, torch.randn(1, 2, 64)).shape == (1, 4, 8)


def test_product_classes_refuse_to_compute_without_a_gpu():
    from trajectorydiffusion_amd import SDUNet_CAR, VAE_InvTrans
    from trajectorydiffusion_amd._lib import TdxError
    x, t, cat, num = small_inputs(2, 4, 8, 50)
    with pytest.raises(TdxError):
        SDUNet_CAR(4, [128, 128], 50, res_blocks=2).eval()(x, t, cat, num)
    with pytest.raises(TdxError):
        VAE_InvTrans().eval().encode(torch.randn(2, 2, 64))
    with pytest.raises(NotImplementedError):
        VAE_InvTrans()(torch.randn(2, 2, 64))


def test_sampler_slice_policy(monkeypatch):
    from trajectorydiffusion_amd.sampler import auto_shards
    monkeypatch.delenv("TDX_SAMPLER_SHARDS", raising=False)
    assert [auto_shards(b, False) for b in (1, 16, 2047, 2048, 4096, 65536)] == [1, 1, 1, 2, 2, 2]
    assert auto_shards(4096, True) == 1 and auto_shards(4097, False) == 1
    monkeypatch.setenv("TDX_SAMPLER_SHARDS", "4")
    assert auto_shards(16, False) == 4
